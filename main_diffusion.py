"""CLI drop-in for the reference's main_diffusion.py:13-25 without absl/ml_collections:

    python main_diffusion.py --config=configs/res64.py --mode=uncond_gen \\
        --config.eval.eval_dir=out --config.eval.ckpt_path=ckpt.pth [--config.eval.batch_size=8]

`--config` may be a reference-style python config file (get_config()) or one of the built-in
names `res64` / `res128`.  `--mode=train` runs one rank per GPU under torchrun.
"""
import sys

from meshdiffusion_amd import config as mdconfig


def parse(argv):
    cfg_path, mode, rest = None, None, []
    i = 0
    while i < len(argv):
        a = argv[i]
        for key in ("--config", "--mode"):
            if a == key or a.startswith(key + "="):
                val = a.split("=", 1)[1] if "=" in a else argv[i + 1]
                i += 0 if "=" in a else 1
                if key == "--config":
                    cfg_path = val
                else:
                    mode = val
                break
        else:
            rest.append(a)
        i += 1
    if cfg_path is None or mode is None:
        raise SystemExit("flags --config and --mode are required")
    if mode not in ("train", "uncond_gen", "cond_gen"):
        raise SystemExit(f"--mode must be one of train|uncond_gen|cond_gen, got {mode}")
    if cfg_path in ("res64", "res128"):
        config = getattr(mdconfig, f"get_config_{cfg_path}")()
    else:
        config = mdconfig.load_config_file(cfg_path)
    left = mdconfig.apply_overrides(config, rest)
    if left:
        raise SystemExit(f"unknown flags: {left}")
    return config, mode


def main(argv=None):
    config, mode = parse(sys.argv[1:] if argv is None else argv)
    from meshdiffusion_amd.lib.diffusion import evaler
    if mode == "uncond_gen":
        evaler.uncond_gen(config)
    elif mode == "cond_gen":
        evaler.cond_gen(config)
    else:
        from meshdiffusion_amd.lib.diffusion import trainer
        trainer.train(config)


if __name__ == "__main__":
    main()
