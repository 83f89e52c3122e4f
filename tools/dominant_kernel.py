"""Launch ONLY the roofline kernel of bench.py (16->16 3x3 conv at 256x256, N=12, bf16, BN-affine
prologue + statistics epilogue) a few times -- the target of the rocprofv3 --pmc passes that give
`roofline.traffic` (FETCH_SIZE / WRITE_SIZE must be collected in separate passes on gfx950)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "3d":            # the 3D bench's roofline kernel: 16->16 3x3x3 at 112x112x80, N = 2
        print(bench.dominant_kernel_roofline_3d(torch.bfloat16, 2, (112, 112, 80)))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
        print(bench.dominant_kernel_roofline(None, torch.bfloat16, n, 256))
