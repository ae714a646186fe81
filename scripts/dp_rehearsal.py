"""Developer helper: one rank of a data-parallel rehearsal of train_dexnerf.py (launch with torch.distributed.run and
DEXNERF_DIST_BACKEND=gloo to put several ranks on one GPU).  DP_EXTRA = further driver arguments (e.g. "--no-hip-graph");
prints one RESULT line per rank: rank, first / last logged training PSNR, milliseconds per iteration over the steady part, HIP graphs replayed per iteration."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import train_dexnerf
args = ["--iters", os.environ.get("DP_ITERS", "150"), "--size", "32", "--views", "6", "--num-random-rays", os.environ.get("DP_RAYS", "512"),
        "--layers", "4", "--width", "128", "--validate-every", "0", "--quiet", "--precision", os.environ.get("DP_PRECISION", "bf16")]
args += os.environ.get("DP_EXTRA", "").split()
args += ["--save", os.path.join(os.environ.get("CKDIR", "/tmp"), f"dp_rank{os.environ.get('RANK', '0')}.ckpt")]
os.environ["DEXNERF_SAVE_ALL_RANKS"] = "1"
res = train_dexnerf.main(args)
# one write() per rank: print() with several arguments issues several writes, and two ranks share the pipe
sys.stdout.write("RESULT %s %r %r %.4f %d\n" % (os.environ.get("RANK", "0"), res["history"][0][2], res["history"][-1][2],
                                             res.get("steady_ms_per_iter", float("nan")), res["hip_graphs"]))
sys.stdout.flush()
