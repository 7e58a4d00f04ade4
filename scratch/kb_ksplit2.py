"""Fine sweep of M around 8 x 1370 rows for the qkv shape: where does the 87 -> 108 us step come from?"""
import os
os.environ["KB_NO_WS"] = "1"
import sys
sys.argv = [sys.argv[0]]
sys.path.insert(0, "scratch")
from kbench import gemm

for M in (10880, 10912, 10944, 10960, 10976, 11008, 11072, 11136, 11264, 11520, 12288, 13056):
    gemm(M, 1152, 384, label=f"{(M + 63) // 64 * 18}t")
for M in (10880, 10960, 11008):
    gemm(M, 384, 1536, label=f"fc2 {(M + 63) // 64 * 6}t")
    gemm(M, 1536, 384, act=1, label=f"fc1 {(M + 63) // 64 * 24}t")
