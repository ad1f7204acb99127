"""Offline tokenisation: images -> code indices with the stage-1 tokenizer (what the reference's stage-2 models consume through
ViTVQ.encode_codes, reference enhancing/modules/stage2/transformer.py:111-113) — SURVEY.md §8f rank 3.

Writes a flat little-endian code file plus a JSON header:
    <out>.codes  : uint16 [n_images, n_tokens (, depth)]   (K <= 65536; int32 otherwise)
    <out>.json   : {"n_images", "n_tokens", "depth", "n_embed", "dtype", "config", "checkpoint"}
and round-trips a sample through decode_codes as a self-check.  Usage:
    python tools/tokenize_dataset.py -c imagenet_vitvq_base --n 512 --batch 64 --out /tmp/codes [--ckpt path]
Data comes from the config's `dataset.params.validation` node (synthetic generator in the shipped yaml)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
from enhancing.utils.general import get_config_from_file, initialize_from_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    cfg = get_config_from_file(os.path.join(ROOT, "configs", args.config + ".yaml"))
    if args.ckpt:
        cfg.model.params["path"] = args.ckpt
    model = initialize_from_config(cfg.model)
    ds = initialize_from_config(cfg.dataset.params.validation if "validation" in cfg.dataset.params else cfg.dataset.params.train)
    q = model.quantizer
    dtype = np.uint16 if q.n_embed <= 65536 else np.int32
    n = min(args.n, len(ds))
    codes_all, t_gpu, first = [], 0.0, None
    with torch.no_grad():
        for s in range(0, n, args.batch):
            x = torch.stack([ds[i]["image"] for i in range(s, min(s + args.batch, n))]).cuda()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            codes = model.encode_codes(x)
            torch.cuda.synchronize(); t_gpu += time.perf_counter() - t0
            if first is None:
                first = (x, codes)
            codes_all.append(codes.cpu().numpy().astype(dtype))
    codes_all = np.concatenate(codes_all)
    codes_all.tofile(args.out + ".codes")
    hdr = dict(n_images=int(codes_all.shape[0]), n_tokens=int(codes_all.shape[1]), depth=int(q.depth), n_embed=int(q.n_embed),
               dtype=np.dtype(dtype).name, config=args.config, checkpoint=args.ckpt)
    json.dump(hdr, open(args.out + ".json", "w"))
    # self-check: file round-trip + decode_codes
    back = np.fromfile(args.out + ".codes", dtype=dtype).reshape(codes_all.shape)
    assert np.array_equal(back, codes_all)
    x, codes = first
    rec = model.decode_codes(torch.from_numpy(back[:x.shape[0]].astype(np.int64)))
    xrec, _ = model(x)
    err = ((rec - xrec).norm() / xrec.norm()).item()
    print(json.dumps(dict(hdr, encode_images_per_s=round(n / t_gpu, 1), decode_codes_vs_forward_rel_err=err)))
    assert err < 2e-2, err


if __name__ == "__main__":
    main()
