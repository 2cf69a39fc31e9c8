"""Dev: p50 of the scene-batch decision (BASELINE config 3: 8 objects x 32768 proxy points x 200 cameras), bench.measure_nbv_batch."""
import os, sys, argparse, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = argparse.Namespace(nbv_iters=int(sys.argv[1]) if len(sys.argv) > 1 else 30, cams=200)
r = bench.measure_nbv_batch(torch.device("cuda:0"), 0, 1, a, variant=int(os.environ["VARIANT"]) if os.environ.get("VARIANT") else None)   # VARIANT=7: the 16-bit path
print("batch p50 ms", r["p50_ms"])
