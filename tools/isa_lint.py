#!/usr/bin/env python3
"""Read-before-wait lint of gfx950 assembly (hipcc -S --cuda-device-only ...): for every kernel, walk the instruction stream in program
order, keep the vector-memory operations in flight (loads with their destination registers; stores count too -- gfx9 has one vmcnt) and the
LDS / scalar-memory reads in flight, retire them at `s_waitcnt vmcnt(N)` / `lgkmcnt(N)`, and report every instruction that READS a register
a load still in flight is going to write.  What it is for: the rotary epilogue of tdf3_kernel returned wrong lanes at random
when hipcc packed its scalar multiplies into v_pk_mul_f32 / v_pk_fma_f32 reading just-loaded table registers -- is a wait missing in front
of the packed operation (a compiler bug this lint would catch in any kernel of the library), or is it there (a hardware hazard no wait
covers)?

    python tools/isa_lint.py file.s [--kernel REGEX] [--show-packed]     exit code 1 when a read-before-wait is found

The walk is linear (labels and branches do not fork the state): a loop's back edge or a join can hide or invent a finding, so a finding is
printed with its line number for a look at the listing.  `--show-packed` also prints, per kernel, every packed-fp32 instruction whose source
is the destination of a vector-memory load of the same kernel, with the wait that covered it."""
import argparse
import re
import sys

REG = re.compile(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b")
WAIT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")
VM_LOAD = re.compile(r"^(global_load_|buffer_load_|flat_load_|scratch_load_)")
VM_STORE = re.compile(r"^(global_store_|buffer_store_|flat_store_|scratch_store_|global_atomic|buffer_atomic|flat_atomic)")
LDS_READ = re.compile(r"^ds_(read|load|bpermute|permute|swizzle|consume|append|ordered)")
LDS_OTHER = re.compile(r"^ds_")
SMEM = re.compile(r"^s_(load|buffer_load)_")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            for i in range(int(m.group(2)), int(m.group(3)) + 1):
                out.add(m.group(1) + str(i))
        else:
            out.add(m.group(4) + m.group(5))
    return out


def lint(lines, show_packed):
    findings, packed = [], []
    vm, lgkm = [], []          # in flight, oldest first: (line number, text, destination registers)
    loaded_by = {}             # register -> line of the vector-memory load that wrote it last (for --show-packed)
    retired_at = {}            # register -> (instruction index, line, text) of the s_waitcnt that retired its load
    last_vm_wait = None
    idx = 0
    for no, raw in lines:
        ins = raw.split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        op, _, rest = ins.partition(" ")
        idx += 1
        if op == "s_waitcnt":
            for name, n in WAIT.findall(rest):
                n = int(n)
                if name == "vmcnt":
                    gone = vm[:len(vm) - n] if n < len(vm) else []
                    for (_, _, ldst) in gone:
                        for r in ldst:
                            retired_at[r] = (idx, no, ins)
                    vm = vm[len(vm) - n:] if n < len(vm) else vm
                    last_vm_wait = (no, ins)
                elif name == "lgkmcnt":
                    lgkm = lgkm[len(lgkm) - n:] if n < len(lgkm) else lgkm
            if "vmcnt" not in rest and "lgkmcnt" not in rest and "expcnt" not in rest:   # s_waitcnt 0 style
                vm, lgkm = [], []
            continue
        if op == "s_barrier" or op.startswith("s_endpgm"):
            continue
        ops = [t.strip() for t in rest.split(",")] if rest else []
        is_lds_dma = "lds" in rest.split() if VM_LOAD.match(op) else False
        if VM_LOAD.match(op) and not is_lds_dma:
            dst, srcs = regs_of(ops[0]) if ops else set(), set().union(*[regs_of(t) for t in ops[1:]]) if len(ops) > 1 else set()
        elif LDS_READ.match(op) or SMEM.match(op):
            dst, srcs = regs_of(ops[0]) if ops else set(), set().union(*[regs_of(t) for t in ops[1:]]) if len(ops) > 1 else set()
        elif VM_STORE.match(op) or LDS_OTHER.match(op) or (VM_LOAD.match(op) and is_lds_dma):
            dst, srcs = set(), set().union(*[regs_of(t) for t in ops]) if ops else set()
        else:
            dst = regs_of(ops[0]) if ops else set()
            srcs = set().union(*[regs_of(t) for t in ops[1:]]) if len(ops) > 1 else set()
            if op.startswith("v_mfma") or op.startswith("v_fma_mix") or op.startswith("v_mov_b32_dpp") or op.startswith("v_cndmask") or "dpp" in rest:
                srcs |= dst                              # read-modify-write forms
        for kind, fl in (("vmcnt", vm), ("lgkmcnt", lgkm)):
            for (lno, ltxt, ldst) in fl:
                hit = (srcs | dst) & ldst               # a write to a register with a load in flight is as wrong as a read
                if hit:
                    findings.append((no, ins, kind, lno, ltxt, sorted(hit)))
        if show_packed and op.startswith("v_pk_") and op.endswith("_f32"):
            src_loaded = [(r, loaded_by[r]) for r in sorted(srcs) if r in loaded_by]
            if src_loaded:
                gaps = [idx - retired_at[r][0] - 1 for r, _ in src_loaded if r in retired_at]
                packed.append((no, ins, src_loaded, last_vm_wait, min(gaps) if gaps else None))
        if VM_LOAD.match(op) and not is_lds_dma:
            vm.append((no, ins, dst))
            for r in dst:
                loaded_by[r] = no
        elif VM_STORE.match(op) or (VM_LOAD.match(op) and is_lds_dma):
            vm.append((no, ins, set()))
        elif LDS_READ.match(op) or SMEM.match(op):
            lgkm.append((no, ins, dst))
        elif LDS_OTHER.match(op):
            lgkm.append((no, ins, set()))
        for r in dst:                                   # any other write ends the "loaded by" record
            if not (VM_LOAD.match(op) and not is_lds_dma):
                loaded_by.pop(r, None)
                retired_at.pop(r, None)
    return findings, packed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default=".*")
    ap.add_argument("--show-packed", action="store_true")
    ap.add_argument("--near", type=int, default=3, help="a packed read this many instructions (or fewer) behind the wait that covered its load is the rotary-epilogue signature")
    ap.add_argument("--quiet", action="store_true", help="print only the kernels with a finding or a near packed read")
    a = ap.parse_args()
    tot_packed = tot_near = 0
    pat = re.compile(a.kernel)
    kernels, cur, name = [], None, None
    with open(a.asm) as fh:
        for no, line in enumerate(fh, 1):
            m = re.match(r"^(_Z\w+):", line)
            if m and ".amdhsa_kernel" not in line:
                name, cur = m.group(1), []
                kernels.append((name, cur))
                continue
            if cur is not None:
                cur.append((no, line.rstrip("\n")))
                if line.strip().startswith("s_endpgm"):
                    cur = None
    bad = 0
    for name, lines in kernels:
        if not pat.search(name):
            continue
        findings, packed = lint(lines, a.show_packed)
        near = [p for p in packed if p[4] is not None and p[4] <= a.near]
        tot_packed += len(packed)
        tot_near += len(near)
        if a.quiet and not findings and not near:
            continue
        print(f"{name}: {len(lines)} lines, {len(findings)} read-before-wait finding(s)" +
              (f", {len(packed)} packed-fp32 reads of loaded registers, {len(near)} of them within {a.near} instructions of the wait that covers the load" if a.show_packed else ""))
        for (no, ins, kind, lno, ltxt, hit) in findings[:20]:
            print(f"   line {no}: `{ins}` touches {hit} while line {lno} `{ltxt}` is in flight ({kind})")
        bad += len(findings)
        for (no, ins, src_loaded, w, gap) in (near if a.quiet else packed)[:12]:
            print(f"   packed, line {no}: `{ins}`; loaded sources {[(r, f'line {ln}') for r, ln in src_loaded[:4]]}; last vmcnt wait: " +
                  (f"line {w[0]} `{w[1]}`" if w else "none") + f"; instructions between the covering wait and this read: {gap}")
    if a.show_packed:
        print(f"total: {tot_packed} packed-fp32 reads of vector-memory-loaded registers, {tot_near} within {a.near} instructions of the covering wait")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
