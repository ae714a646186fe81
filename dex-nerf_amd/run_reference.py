#!/usr/bin/env python3
"""Run one of the reference's own scripts (train_dexnerf_rgb.py, train_dexnerf_ir.py, train_nerf*.py, eval_nerf.py,
tiny_nerf.py, cache_dataset.py) against THIS build's `nerf` package, unmodified:

    python /path/to/repo/dex-nerf_amd/run_reference.py /path/to/Dex-NERF/nerf-pytorch/train_dexnerf_rgb.py \
           --config config/messytable-obj-edward.yml

Why a launcher and not PYTHONPATH: `python script.py` puts the script's directory at sys.path[0], AHEAD of PYTHONPATH, and
that directory holds the reference's own `nerf/` (train_dexnerf_rgb.py:15-19 does `from nerf import ...`), so a PYTHONPATH
swap silently keeps running the reference's CPU/torch path.  Here the build's package directory is inserted at sys.path[0]
and the script runs through runpy.run_path, which - unlike `python script.py` - does not add the script's directory; that
directory is appended BEHIND ours so the script's sibling modules (its own helpers, `lieutils`, configs by relative path ...)
still resolve.  The working directory becomes the script's, as the reference's relative config/log paths assume.
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit(f"run_reference: no such script: {script}")
    script_dir = os.path.dirname(script)
    # ours first; the script's directory behind it (never in front: that is the whole point)
    sys.path[:] = [HERE] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (HERE, script_dir)] + [script_dir]
    for name in [m for m in sys.modules if m == "nerf" or m.startswith("nerf.")]:
        del sys.modules[name]
    import nerf
    if os.path.dirname(os.path.dirname(os.path.abspath(nerf.__file__))) != HERE:
        raise SystemExit(f"run_reference: `nerf` resolved to {nerf.__file__}, not to the build's package under {HERE}")
    from nerf import _hip
    _hip.lib()   # fail here, loudly, if libdexnerf_hip.so is missing - not at the first render
    precision = os.environ.get("DEXNERF_PRECISION")
    if precision:
        nerf.set_precision(precision)
    os.chdir(script_dir)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
