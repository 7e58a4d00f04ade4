"""Disassembly helpers for the ISA-level tests (CPU only: llvm-objdump on the gfx950 code objects inside the built library).

The rule these tests enforce was written down in round 2 after a race that 419 green GPU tests could not see: a raw
``s_barrier`` that releases an LDS stage must be preceded by waits on BOTH counters -- ``vmcnt(0)`` (this wave's LDS-DMAs have landed) and
``lgkmcnt(0)`` (this wave's own LDS reads have returned) -- because after the barrier other waves refill the stage."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile
from dataclasses import dataclass
from typing import Dict, List, Optional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "endodav_amd", "lib", "libendodav_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@dataclass
class Inst:
    addr: int
    op: str
    args: str


_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def disassemble(lib: str = LIB) -> Dict[str, List[Inst]]:
    """{mangled kernel name: instructions} over every gfx950 bundle of the library."""
    if not os.path.exists(OBJDUMP):
        raise FileNotFoundError(OBJDUMP)
    out: Dict[str, List[Inst]] = {}
    tmp = tempfile.mkdtemp(prefix="edv_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            text = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur: Optional[List[Inst]] = None
            for line in text.splitlines():
                m = _FUNC.match(line)
                if m:
                    cur = out.setdefault(m.group(1), [])
                    continue
                m = _LINE.match(line)
                if m and cur is not None:
                    cur.append(Inst(int(m.group(3), 16), m.group(1), m.group(2)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def branch_target(i: Inst) -> Optional[int]:
    if not (i.op.startswith("s_cbranch") or i.op == "s_branch"):
        return None
    simm = int(i.args.split()[0])
    if simm >= 32768:
        simm -= 65536
    return i.addr + 4 + 4 * simm


def loop_ranges(insts: List[Inst]):
    """[lo, hi] address ranges of backward branches (loops)."""
    r = []
    for i in insts:
        t = branch_target(i)
        if t is not None and t <= i.addr:
            r.append((t, i.addr))
    return r


def is_lds_access(i: Inst) -> bool:
    return i.op.startswith("ds_")


def is_vmem(i: Inst) -> bool:
    return i.op.startswith(("buffer_", "global_", "flat_", "scratch_"))


def is_lds_dma(i: Inst) -> bool:
    return is_vmem(i) and (re.search(r"\blds\b", i.args) is not None or "_lds_" in i.op)


def waits(i: Inst):
    """(vmcnt, lgkmcnt) an s_waitcnt enforces; None = not constrained by this instruction."""
    if i.op != "s_waitcnt":
        return None, None
    v = re.search(r"vmcnt\((\d+)\)", i.args)
    l = re.search(r"lgkmcnt\((\d+)\)", i.args)
    return (int(v.group(1)) if v else None), (int(l.group(1)) if l else None)


def barrier_violations(insts: List[Inst], loops_only: bool = True, ring: bool = False) -> List[str]:
    """Every s_barrier (inside a loop when `loops_only`) must find, scanning backwards to the previous barrier, a vector-memory wait that leaves no
    LDS-DMA of an EARLIER stretch in flight, and an `s_waitcnt lgkmcnt(0)` with no LDS instruction after it.  The memory wait is either `vmcnt(0)` with no
    LDS-DMA after it (the two-stage kernels: the only form accepted unless `ring`), or, for `ring` kernels, a counted `vmcnt(N)`: the N youngest vector-memory instructions before it may then still be in flight,
    and those may include LDS-DMAs of the CURRENT stretch only (gemm_x6's three-stage ring: the DMA issued in a step fills the stage read two barriers
    later; what must have landed is the DMA of the step before).  A stretch that issues no LDS-DMA and inherits none needs no vmcnt wait, one that issues
    no LDS access needs no lgkmcnt wait.  The scan is linear in address order: the compiler keeps the k-loop bodies of these kernels straight-line."""
    loops = loop_ranges(insts)
    bad = []
    for n, i in enumerate(insts):
        if i.op != "s_barrier":
            continue
        if loops_only and not any(lo <= i.addr <= hi for lo, hi in loops):
            continue
        need_l = True
        dirty_l = False
        wait_at, wait_n = None, None  # the last vmcnt wait of the stretch
        prev_barrier = -1
        for j in range(n - 1, -1, -1):
            p = insts[j]
            if p.op == "s_barrier":
                prev_barrier = j
                break
            v, l = waits(p)
            if need_l and l == 0:
                need_l = False
            if need_l and is_lds_access(p):
                dirty_l = True
            if wait_at is None and v is not None:
                wait_at, wait_n = j, v
        # vector-memory side
        stretch = range(prev_barrier + 1, n)
        dma_after_wait = any(is_lds_dma(insts[j]) for j in stretch if wait_at is None or j > wait_at)
        if wait_at is None:
            if any(is_lds_dma(insts[j]) for j in stretch):
                bad.append(f"{i.addr:#x}: LDS-DMA in flight across s_barrier (no s_waitcnt vmcnt after it)")
        elif not ring:
            if wait_n != 0 and any(is_lds_dma(insts[j]) for j in stretch):
                bad.append(f"{i.addr:#x}: LDS-DMA in flight across s_barrier (a counted s_waitcnt vmcnt({wait_n}) is not a drain)")
            elif dma_after_wait:
                bad.append(f"{i.addr:#x}: LDS-DMA in flight across s_barrier (issued after the s_waitcnt vmcnt(0))")
        else:
            if wait_n == 0 and dma_after_wait:
                bad.append(f"{i.addr:#x}: LDS-DMA in flight across s_barrier (issued after the s_waitcnt vmcnt(0))")
            # the wait_n youngest vector-memory instructions before the wait may be in flight: none of them may be an LDS-DMA of an earlier stretch
            left = wait_n
            for j in range(wait_at - 1, -1, -1):
                if left == 0:
                    break
                p = insts[j]
                if is_vmem(p):
                    left -= 1
                    if is_lds_dma(p) and j < prev_barrier:
                        bad.append(f"{i.addr:#x}: s_waitcnt vmcnt({wait_n}) leaves the LDS-DMA at {p.addr:#x} (an earlier stretch) in flight across s_barrier")
                        break
        if dirty_l:
            bad.append(f"{i.addr:#x}: LDS access in flight across s_barrier (no s_waitcnt lgkmcnt(0) after it)")
    return bad


_VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def _vregs(tok: str):
    m = _VREG.match(tok.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def is_valu(i: Inst) -> bool:
    return i.op.startswith("v_") and not i.op.startswith(("v_mfma", "v_smfma", "v_accvgpr"))


def store_hazard_violations(insts: List[Inst]) -> List[str]:
    """A vector-memory store of more than 64 bits that takes its soffset from an SGPR reads its data registers late: a VALU write of one of them in
    the next two wait states lands in the stored data (scratch/ubench/store_hazard.hip: word 0 of lanes 12..15 of each 16).  LLVM pads the immediate
    form itself but not the register form, so a hand-written one must carry `s_nop` (or two non-VALU instructions) after it."""
    bad = []
    for n, i in enumerate(insts):
        if not (i.op.startswith("buffer_store_dwordx3") or i.op.startswith("buffer_store_dwordx4")):
            continue
        ops = [a.strip() for a in i.args.split(",")]
        if len(ops) < 4 or not ops[3].split()[0].startswith("s"):
            continue
        data = _vregs(ops[0])
        slots = 0
        for p in insts[n + 1:n + 4]:
            if p.op == "s_nop":
                slots += int(p.args.split()[0]) + 1
            else:
                if is_valu(p) and (_vregs(p.args.split(",")[0]) & data):
                    bad.append(f"{p.addr:#x}: {p.op} writes {p.args.split(',')[0]} {slots} wait state(s) after {i.op} {i.args}")
                slots += 1
            if slots >= 2:
                break
    return bad


def reads_after_last_mfma(insts: List[Inst]) -> List[str]:
    """For the multi-stage LDS rings of the bf16 x 6 kernels: in every stretch between two s_barriers that multiplies (contains MFMAs), no LDS READ may
    follow the last MFMA.  A read there belongs to the NEXT stretch -- hoisted above the waits and the barrier that publish the stage it reads (other waves'
    DMAs / plane writes); the machine scheduler did exactly that once the schedule of gemm_x6's steps was spelled out with sched_group_barrier, the inline
    asm's memory clobber notwithstanding."""
    bad = []
    start = 0
    for n, i in enumerate(insts):
        if i.op != "s_barrier":
            continue
        seg = insts[start:n]
        start = n + 1
        last = max((k for k, p in enumerate(seg) if p.op.startswith("v_mfma")), default=None)
        if last is None:
            continue
        for p in seg[last + 1:]:
            if p.op.startswith("ds_read"):
                bad.append(f"{p.addr:#x}: {p.op} after the last MFMA before the s_barrier at {i.addr:#x}")
                break
    return bad
