"""Occupancy timeline of the three MFMA attention kernels from a -DATTN_TRACE build (tools/variant.sh trace attention_mfma.hip -DATTN_TRACE;
run with SSI_HIP_LIB=$PWD/variants/libssi_trace.so): every workgroup's start / end on the 100 MHz clock, the CU it ran on and its tile
count.  Prints, per kernel: makespan against the launch's event time, the fit  duration = a + b x tiles  over the workgroups (a = what a
workgroup costs before and after its tile loop), how many workgroups a CU ran at once, and the idle share of the CU-slots."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, 'speech-integration_amd')
from ssi import _lib, ops  # noqa: E402

B, S, H, KV, hd = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (8, 2048, 32, 8, 64)
T = B * S
torch.manual_seed(0)
qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
dout = torch.randn(T, H * hd, device='cuda').bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty_like(lse)
lib = _lib.load()
fn = lib.ssi_debug_attn_trace
fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int


def timed(f, iters=5):
    for _ in range(3):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


t_fwd = timed(lambda: ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd))
t_bwd = timed(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
torch.cuda.synchronize()
print(f"B={B} S={S} H={H} KV={KV}: forward {t_fwd:.1f} us, backward (dQ + dK/dV) {t_bwd:.1f} us (event times, traced build)")
grids = {0: B * KV * (S // 32), 1: B * KV * (S // 32), 2: B * KV * (S // 128)}
for k, name in ((0, "attn_fwd"), (1, "attn_bwd_dq"), (2, "attn_bwd_dkv")):
    buf = np.zeros((8192, 6), dtype=np.uint64)
    rc = fn(buf.ctypes.data, k)
    assert rc == 0, rc
    n = min(grids[k], 8192)
    tr = buf[:n]
    t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
    work = tr[:, 3].astype(np.float64)
    hw = tr[:, 2]
    xcc, hwid = (hw >> np.uint64(32)).astype(np.int64) & 0xF, (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 0x1, (hwid >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    dur = (t1 - t0) * 0.01  # us
    start = (t0 - t0.min()) * 0.01
    end = (t1 - t0.min()) * 0.01
    span = end.max()
    A = np.stack([np.ones_like(work), work], 1)
    (a, b), *_ = np.linalg.lstsq(A, dur, rcond=None)
    ncu = len(np.unique(cuid))
    busy = dur.sum()
    # concurrency per CU: sample the timeline
    grid_t = np.linspace(0, span, 400)
    conc = np.array([((start <= t) & (end > t)).sum() for t in grid_t]) / max(ncu, 1)
    print(f"{name}: {n} workgroups on {ncu} CUs; first start -> last end {span:.1f} us; sum of workgroup durations / (CUs x span) = "
          f"{busy / (ncu * span):.2f} resident workgroups per CU on average; duration = {a:.2f} us + {b:.3f} us x tiles "
          f"(tiles {work.min():.0f}..{work.max():.0f}, mean {work.mean():.1f}); total tiles {work.sum():.0f}")
    q = [0.0, 0.25, 0.5, 0.75, 0.9, 1.0]
    print("   resident workgroups per CU along the launch: " + ", ".join(f"{int(100 * x)}%: {conc[min(int(x * 399), 399)]:.2f}" for x in q))
    late = start > 0.02 * span
    print(f"   workgroups started after the first 2 % of the launch: {int(late.sum())}; last start at {start.max():.1f} us; "
          f"shortest / longest workgroup {dur.min():.1f} / {dur.max():.1f} us")
    ta, tb = tr[:, 4].astype(np.int64), tr[:, 5].astype(np.int64)
    pro, loop, epi = (ta - t0) * 0.01, (tb - ta) * 0.01, (t1 - tb) * 0.01
    (la, lb), *_ = np.linalg.lstsq(A, loop, rcond=None)
    print(f"   wave 0 of a workgroup: start -> tile loop {pro.mean():.2f} us (max {pro.max():.2f}); tile loop mean {loop.mean():.2f} us = {la:.2f} us + {lb:.3f} us x tiles; "
          f"loop end -> workgroup end {epi.mean():.2f} us (max {epi.max():.2f})")
    if name == "attn_bwd_dkv" or "-v" in sys.argv:  # by work class: when do the workgroups of each size start and end?
        for wv in sorted(set(work.tolist()), reverse=True)[:20]:
            m = work == wv
            print(f"   {int(wv):4d} steps: {int(m.sum()):4d} wgs, start mean {start[m].mean():6.1f} (max {start[m].max():6.1f}), end mean {end[m].mean():6.1f} "
                  f"(max {end[m].max():6.1f}), duration mean {dur[m].mean():6.1f} (min {dur[m].min():6.1f}, max {dur[m].max():6.1f}); per step {1e3 * (dur[m].mean() / wv):.0f} ns")
        # per XCD: end of its last workgroup
        print("   last end per XCD: " + ", ".join(f"{x}: {end[xcc == x].max():.0f}" for x in sorted(set(xcc.tolist()))))
    # per-tile cost by workgroup size class
    for lo, hi in ((1, 4), (5, 12), (13, 24), (25, 400)):
        m = (work >= lo) & (work <= hi)
        if m.any():
            print(f"   tiles {lo:3d}..{hi:3d}: {int(m.sum()):5d} workgroups, mean duration {dur[m].mean():6.2f} us, per tile {dur[m].sum() / work[m].sum():.3f} us")
