"""Training entry with the reference's CLI (reference main.py:17-27): python main.py -c imagenet_vitvq_base -ng 8

Same flags and semantics (-c -s -nn -ng -u -e -lr -a -b -m); `pl.Trainer.fit` is replaced by the in-repo loop
(enhancing.engine.trainer.Trainer).  With -ng > 1 the script re-executes itself under torch.distributed.run, one
process per GPU (what Lightning's DDP strategy does), gradients all-reduced with RCCL over xGMI."""
import argparse
import os
import subprocess
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "enhancing-transformers_amd"))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', type=str, required=True)
    parser.add_argument('-s', '--seed', type=int, default=0)
    parser.add_argument('-nn', '--num_nodes', type=int, default=1)
    parser.add_argument('-ng', '--num_gpus', type=int, default=1)
    parser.add_argument('-u', '--update_every', type=int, default=1)
    parser.add_argument('-e', '--epochs', type=int, default=100)
    parser.add_argument('-lr', '--base_lr', type=float, default=4.5e-6)
    parser.add_argument('-a', '--use_amp', default=False, action='store_true')
    parser.add_argument('-b', '--batch_frequency', type=int, default=750)
    parser.add_argument('-m', '--max_images', type=int, default=4)
    parser.add_argument('--max_steps', type=int, default=None, help="(extension) stop after this many optimizer steps")
    parser.add_argument('--fp32', default=False, action='store_true',
                        help="(extension) run the fp32 'exact' engine mode = Lightning precision 32; cannot be combined with --use_amp")
    parser.add_argument('--bf16', default=False, action='store_true',
                        help="(extension) mixed precision with bf16 MFMA operands (Lightning precision 'bf16': no loss scale, ~3 %% faster, ~5e-3 parity) "
                             "instead of the fp16 operands of --use_amp")
    args = parser.parse_args()

    if args.num_gpus > 1 and "RANK" not in os.environ:
        if args.num_nodes > 1:
            sys.exit("multi-node: launch one torchrun per node yourself (MASTER_ADDR/MASTER_PORT/NODE_RANK)")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.num_gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    from enhancing.engine.trainer import Trainer
    from enhancing.utils.general import AttrDict, get_config_from_file, initialize_from_config, set_seed, setup_callbacks

    set_seed(args.seed)
    config = get_config_from_file(Path(ROOT) / "configs" / (args.config + ".yaml"))
    model = initialize_from_config(config.model)
    model.learning_rate = args.base_lr
    data = initialize_from_config(config.dataset)
    data.prepare_data()
    # reference main.py:52: precision = 16 if --use_amp else 32, i.e. fp16 autocast + GradScaler under --use_amp.  Here that mixed precision (fp16 MFMA
    # operands, fp32 master weights / accumulation, loss-scaled backward with the inf / nan step skip) is the product path and therefore ALSO the
    # default; --use_amp selects it explicitly, --bf16 swaps the operand format, --fp32 selects the fp32 exact mode (what the reference runs without
    # --use_amp), and contradictory combinations are an error instead of being silently resolved
    if args.fp32 and (args.use_amp or args.bf16):
        sys.exit("--fp32 (no mixed precision) contradicts --use_amp / --bf16 (mixed precision)")
    exp_config = AttrDict(vars(args))                 # reference main.py:37: the experiment config is the parsed command line (+ name)
    exp_config.update(name=args.config, epochs=args.epochs, update_every=args.update_every, base_lr=args.base_lr, use_amp=args.use_amp,
                      batch_frequency=args.batch_frequency, max_images=args.max_images)
    callbacks, _logger = setup_callbacks(exp_config, config)      # reference main.py:47
    trainer = Trainer(callbacks=callbacks, max_epochs=args.epochs, precision=32 if args.fp32 else ("bf16" if args.bf16 else 16), gpus=args.num_gpus, num_nodes=args.num_nodes,
                      strategy="ddp" if args.num_nodes > 1 or args.num_gpus > 1 else None, accumulate_grad_batches=args.update_every,
                      max_steps=args.max_steps, default_root_dir=os.path.join(ROOT, "experiments", args.config))
    trainer.fit(model, data)
