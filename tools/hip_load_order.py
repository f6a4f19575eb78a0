#!/usr/bin/env python3
"""Does it matter whether libdvo_hip.so or PyTorch initialises HIP first in a process?  (It did: PyTorch-ROCm loads its own
libamdhip64 by path, and with the system copy already in the process torch.cuda then finds no GPU; dvo_amd._cabi.load_library
therefore imports torch before it loads the extension.)
    python tools/hip_load_order.py ctx_first | torch_first"""
import sys


def main(order):
    sys.path.insert(0, "dense-visual-odometry_amd")
    import dvo_amd
    if order == "torch_first":
        import torch
        print("torch sees", torch.cuda.device_count(), torch.zeros(1, device="cuda").item())
    ctx = dvo_amd.HipContext(max_w=64, max_h=48, max_levels=1, n_slots=2, max_pairs=1)
    print("ctx ok", ctx is not None)
    import torch
    print("torch after ctx: count", torch.cuda.device_count())
    try:
        print(torch.zeros(1, device="cuda").item(), "torch cuda ok after ctx")
    except Exception as e:
        print("FAILED:", repr(e)[:200])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ctx_first")
