#!/usr/bin/env python3
"""Listing lint of the built library (no GPU needed): the hazards of the hand-pipelined kernels that only a look at the
disassembly used to verify (profiles/LAB_NOTES.md, round 4), checked mechanically on the gfx950 code objects inside
``libssi_hip.so``.

Rules
  R1  no scratch: kernels named in ``NO_SCRATCH`` have private_segment_fixed_size == 0 and no ``scratch_*`` instruction;
  R2  main loops keep their registers: a basic block that loops to itself and holds >= ``MAIN_LOOP_MFMAS`` matrix instructions
      (the one-basic-block trips of the pipelined kernels) contains no ``v_accvgpr_mov / _write / _read`` — hipcc's copies of
      accumulation registers between inline-asm MFMAs get no wait states (a dQ element wrong by 1.2e-1, no fault);
  R3  nothing in flight towards LDS at the end: in every kernel that issues LDS-DMA (``buffer_load ... lds``,
      ``global_load_lds_*``) every path from such an instruction to ``s_endpgm`` passes ``s_waitcnt vmcnt(0)`` — the LDS belongs
      to the next workgroup of the CU by then (the round-4 race of attn_bwd_dq2_kernel).

``python tools/kernel_lint.py [lib]`` prints one line per kernel and exits 1 on a violation; ``tests/test_kernel_listing.py``
runs the same checks in the CPU suite."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from dataclasses import dataclass, field

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "speech-integration_amd", "libssi_hip.so")
NO_SCRATCH = ("attn_bwd_dq2_kernel", "attn_bwd_dkv2_kernel", "attn_fwd_kernel", "gemm_nt4dma_kernel")
PINNED_LOOPS = ("attn_bwd_dq2_kernel", "attn_bwd_dkv2_kernel")   # kernels whose main loops are inline-asm MFMAs on pinned register classes
MAIN_LOOP_MFMAS = 32

_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_SYM = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_TGT = re.compile(r"<(\S+?)\+0x([0-9a-f]+)>\s*$")
_TGT0 = re.compile(r"<(\S+?)>\s*$")


@dataclass
class Ins:
    addr: int
    op: str
    args: str
    target: int | None = None


@dataclass
class Kernel:
    name: str
    addr: int
    ins: list[Ins] = field(default_factory=list)
    scratch_bytes: int = 0


def extract(lib: str, workdir: str) -> list[str]:
    """Code objects of the fat binary: llvm-objdump --offloading writes one file per translation unit next to its input."""
    link = os.path.join(workdir, "lib.so")
    os.symlink(os.path.abspath(lib), link)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], check=True, cwd=workdir, stdout=subprocess.DEVNULL)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if f.endswith("gfx950"))


def scratch_sizes(co: str) -> dict[str, int]:
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, name = {}, None
    for line in notes.splitlines():   # kernel-level keys are sorted: .name directly in front of .private_segment_fixed_size
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m2 = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
        if m2 and name:
            out[name] = int(m2.group(1))
    return out


def disassemble(co: str) -> list[Kernel]:
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    syms, kernels, cur = {}, [], None
    lines = text.splitlines()
    for line in lines:
        m = _SYM.match(line)
        if m:
            syms[m.group(2)] = int(m.group(1), 16)
    for line in lines:
        m = _SYM.match(line)
        if m:
            cur = Kernel(m.group(2), int(m.group(1), 16))
            kernels.append(cur)
            continue
        m = _INS.match(line)
        if not m or cur is None:
            continue
        ins = Ins(int(m.group(3), 16), m.group(1), m.group(2))
        if ins.op.startswith(("s_cbranch", "s_branch")):
            t = _TGT.search(line)
            if t:
                ins.target = syms[t.group(1)] + int(t.group(2), 16)
            else:
                t0 = _TGT0.search(line)
                if t0 and t0.group(1) in syms:
                    ins.target = syms[t0.group(1)]
        cur.ins.append(ins)
    return [k for k in kernels if k.ins]


def blocks(k: Kernel) -> list[tuple[int, int]]:
    """Basic blocks as [first, last] instruction indices."""
    leaders = {0}
    index = {i.addr: n for n, i in enumerate(k.ins)}
    for n, i in enumerate(k.ins):
        if i.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            if n + 1 < len(k.ins):
                leaders.add(n + 1)
            if i.target is not None and i.target in index:
                leaders.add(index[i.target])
    ls = sorted(leaders)
    return [(a, (ls[j + 1] - 1) if j + 1 < len(ls) else len(k.ins) - 1) for j, a in enumerate(ls)]


def is_lds_dma(i: Ins) -> bool:
    return i.op.startswith("global_load_lds") or (i.op.startswith("buffer_load") and re.search(r"\blds\b", i.args) is not None)


def drains_vm(i: Ins) -> bool:
    if i.op != "s_waitcnt":
        return False
    if re.search(r"vmcnt\(0\)", i.args):
        return True
    return re.fullmatch(r"(0x)?0+", i.args.strip()) is not None   # raw immediate 0: every counter


def lint_kernel(k: Kernel) -> list[str]:
    errs = []
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+|^_Z\d+", "", k.name)
    if any(s in k.name for s in NO_SCRATCH):
        if k.scratch_bytes:
            errs.append(f"R1 {short}: {k.scratch_bytes} bytes of scratch")
        n = sum(1 for i in k.ins if i.op.startswith("scratch_"))
        if n:
            errs.append(f"R1 {short}: {n} scratch instructions")
    bl = blocks(k)
    index = {i.addr: n for n, i in enumerate(k.ins)}
    if any(s in k.name for s in PINNED_LOOPS):
        loops = 0
        for a, b in bl:
            last = k.ins[b]
            if last.target is None or index.get(last.target) != a:
                continue
            body = k.ins[a:b + 1]
            if sum(1 for i in body if i.op.startswith("v_mfma")) < MAIN_LOOP_MFMAS:
                continue
            loops += 1
            bad = [i for i in body if i.op.startswith(("v_accvgpr_mov", "v_accvgpr_write", "v_accvgpr_read"))]
            if bad:
                errs.append(f"R2 {short}: {len(bad)} accumulation-register copies inside the main loop at {k.ins[a].addr:#x}")
        if loops == 0:
            errs.append(f"R2 {short}: no one-basic-block main loop found (a trip was split: check the listing)")
    if any(is_lds_dma(i) for i in k.ins):
        # forward may-analysis over the CFG: dirty = an LDS-DMA request may be in flight
        succ: dict[int, list[int]] = {}
        start_of = {a: n for n, (a, _) in enumerate(bl)}
        for n, (a, b) in enumerate(bl):
            last, s = k.ins[b], []
            if last.op.startswith("s_endpgm"):
                pass
            elif last.op.startswith("s_branch"):
                if last.target in index:
                    s.append(start_of[index[last.target]])
            else:
                if last.op.startswith("s_cbranch") and last.target in index:
                    s.append(start_of[index[last.target]])
                if n + 1 < len(bl):
                    s.append(n + 1)
            succ[n] = s
        def transfer(n: int, dirty: bool, report: bool = False) -> bool:
            a, b = bl[n]
            for i in k.ins[a:b + 1]:
                if is_lds_dma(i):
                    dirty = True
                elif drains_vm(i):
                    dirty = False
                elif report and dirty and i.op.startswith("s_endpgm"):
                    errs.append(f"R3 {short}: s_endpgm at {i.addr:#x} reachable with LDS-DMA requests in flight (no s_waitcnt vmcnt(0) on the path)")
            return dirty

        dirty_in = [False] * len(bl)
        work = list(range(len(bl)))
        while work:  # to the fixed point: a block is dirty on entry if any predecessor may leave it dirty
            n = work.pop()
            if transfer(n, dirty_in[n]):
                for s_ in succ[n]:
                    if not dirty_in[s_]:
                        dirty_in[s_] = True
                        work.append(s_)
        for n in range(len(bl)):
            transfer(n, dirty_in[n], report=True)
    return errs


def lint(lib: str = DEFAULT_LIB) -> tuple[list[str], dict[str, dict]]:
    errs, report = [], {}
    with tempfile.TemporaryDirectory() as wd:
        for co in extract(lib, wd):
            sizes = scratch_sizes(co)
            for k in disassemble(co):
                k.scratch_bytes = sizes.get(k.name, 0)
                e = lint_kernel(k)
                errs += e
                report[k.name] = {"instructions": len(k.ins), "scratch": k.scratch_bytes, "mfma": sum(1 for i in k.ins if i.op.startswith("v_mfma")),
                                  "lds_dma": sum(1 for i in k.ins if is_lds_dma(i)), "errors": e}
    return errs, report


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    errs, report = lint(lib)
    for name, r in sorted(report.items()):
        if r["mfma"] or r["lds_dma"] or r["scratch"]:
            print(f"{'FAIL' if r['errors'] else 'ok  '} {name[:110]:110s} ins {r['instructions']:6d} mfma {r['mfma']:4d} lds-dma {r['lds_dma']:3d} scratch {r['scratch']}")
    for e in errs:
        print("VIOLATION", e)
    sys.exit(1 if errs else 0)
