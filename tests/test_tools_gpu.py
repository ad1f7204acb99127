"""Driver-run coverage of the command-line entry points around the hot path: the offline tokenizer and the
multi-process bench line (§8e) — the latter as two ranks sharing this GPU over gloo, so the bucket accounting of the real base-size module tree is
checked on hardware even though no multi-GPU node (RCCL) is available to the build."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokenize_dataset_encode_file_decode_round_trip(tmp_path):
    """tools/tokenize_dataset.py: images -> encode_codes -> uint16 code file + header -> read back -> decode_codes == the model's own reconstruction
    (reference vitvqgan.py:74-90: what the stage-2 transformers consume)."""
    out = str(tmp_path / "codes")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tokenize_dataset.py"), "-c", "imagenet_vitvq_small", "--n", "16", "--batch", "8",
                        "--out", out], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    hdr = json.load(open(out + ".json"))
    assert hdr["n_images"] == 16 and hdr["n_tokens"] == 1024 and hdr["depth"] == 1 and hdr["n_embed"] == 8192 and hdr["dtype"] == "uint16"
    codes = np.fromfile(out + ".codes", dtype=np.uint16).reshape(16, 1024)
    assert codes.max() < 8192
    assert line["decode_codes_vs_forward_rel_err"] < 2e-2
    # an independent decode of the FILE: a fresh model (same seed-0 init as the tool's) reconstructs from the stored codes
    sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
    from enhancing.utils.general import get_config_from_file, initialize_from_config
    cfg = get_config_from_file(os.path.join(ROOT, "configs", "imagenet_vitvq_small.yaml"))
    model = initialize_from_config(cfg.model)
    rec = model.decode_codes(torch.from_numpy(codes[:4].astype(np.int64)))
    assert rec.shape == (4, 3, 256, 256) and torch.isfinite(rec).all()


def test_bench_two_ranks_on_one_gpu_over_gloo_accounts_for_every_gradient_byte():
    """`bench.py --gpus 2` as torch.distributed.run launches it, with the two ranks sharing cuda:0 over gloo (RCCL refuses two ranks per device): at the
    BASE config every parameter's gradient must travel exactly once per step (bytes_reduced == 4 * n_params, nothing left to the safety net), and the
    line carries the per-rank exposed-communication block."""
    env = dict(os.environ, ENH_DIST_BACKEND="gloo", ENH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    comm = line["comm"]
    assert comm["backend"] == "gloo" and len(comm["per_rank"]) == 2
    n = comm["n_params"]
    assert 170.0e6 < n < 171.5e6          # 170.66 M trainable + 64-element alignment padding of the flat buffer
    for pr in comm["per_rank"]:
        assert pr["gap_elems"] == 0
        assert pr["bytes_reduced_per_step"] == 4 * n, (pr, n)
        assert pr["host_ms"] is not None and pr["stream_ms"] is not None and pr["buckets_per_step"] >= 20
    assert line["roofline"]["frac"] > 0 and line["value"] > 0


def _json_lines(stdout):
    return [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]


def test_bench_bare_single_gpu_form_prints_one_json_line():
    """the driver's N = 1 command, `python bench.py --gpus 1 ...` with no launcher environment: exactly ONE JSON line on stdout"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
                        "--no-parity-mode"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0


def test_bench_bare_multi_gpu_form_launches_itself_two_ranks_over_gloo():
    """VERDICT r4 missing #1: `python bench.py --gpus 2` with NO launcher environment (the form of the driver's N = 1 command) must not die on the
    WORLD_SIZE assert: it re-executes itself under torch.distributed.run.  Here the two ranks share cuda:0 over gloo (one-GPU box); the RCCL form of the
    same command runs in the next test wherever two GPUs exist."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")}
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env.update(ENH_DIST_BACKEND="gloo", ENH_FORCE_DEVICE="0", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
                        "--no-parity-mode"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["comm"]["backend"] == "gloo"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks per device)")
def test_bench_bare_multi_gpu_form_over_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "ENH_DIST_BACKEND", "ENH_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity-mode"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["comm"]["backend"] == "nccl" and line["value"] > 0


def test_main_py_cli_trains_and_checkpoints(tmp_path):
    """north_star: "Lightning-style main.py entry so it drops in for the stage-1 tokenizer loop".  The reference's command line (main.py:17-27) on the small
    config with the synthetic dataset: a few optimizer steps through Trainer.fit on the GPU, the loss is finite and moves, a Lightning-format checkpoint
    ({"state_dict": ...} with the reference's keys) is written and loads back into a fresh model."""
    import glob
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "-c", "imagenet_vitvq_small", "-e", "1", "-lr", "1e-4", "--max_steps", "4", "-u", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rows = [json.loads(ln) for ln in open(os.path.join(ROOT, "experiments", "imagenet_vitvq_small", "metrics.jsonl"))]
    val = [row for row in rows if "val/rec_loss" in row]
    assert val and np.isfinite(val[-1]["val/rec_loss"]) and np.isfinite(val[-1]["val/total_loss"])          # the epoch-end validation pass ran on the trained weights
    ck = sorted(glob.glob(os.path.join(ROOT, "experiments", "imagenet_vitvq_small", "ckpt", "*.ckpt")))
    assert ck, "no checkpoint written"
    sd = torch.load(ck[-1], map_location="cpu")
    assert "state_dict" in sd and "encoder.transformer.layers.0.0.fn.to_qkv.weight" in sd["state_dict"] and sd["global_step"] >= 4
    sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))
    from enhancing.utils.general import get_config_from_file, initialize_from_config
    model = initialize_from_config(get_config_from_file(os.path.join(ROOT, "configs", "imagenet_vitvq_small.yaml")).model)
    missing = model.load_state_dict(sd["state_dict"], strict=False)
    assert not [k for k in missing.missing_keys if not k.startswith("loss.")], missing.missing_keys[:5]
