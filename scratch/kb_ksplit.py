"""Same FLOPs, different K depth per tile (what a 2- or 4-way K split of every tile would give before its fix-up cost), and a
grid-size sweep at fixed K: time = fixed + per-round (ramp/tail vs steady state of the 64x64 LDS-DMA GEMM)."""
import os
os.environ["KB_NO_WS"] = "1"
import sys
sys.argv = [sys.argv[0]]
sys.path.insert(0, "scratch")
from kbench import gemm

M = 8 * 1370
for (N, K) in ((1152, 384), (2304, 192), (4608, 96), (384, 1536), (768, 768), (1536, 384), (3072, 192)):
    gemm(M, N, K, label="iso-flops")
print("--- grid sweep, N=1152 K=384 (18 column tiles): rows -> tiles")
for tiles in (256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 6144, 8192):
    rows = tiles // 18 * 64
    gemm(rows, 1152, 384, label=f"{rows // 64 * 18}t")
