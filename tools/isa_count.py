#!/usr/bin/env python3
"""Static facts about the dominant kernel, from the library that was actually built: disassembles `k_reduce<V1, none, fast>` out of
libdvo_hip.so (llvm-objdump on the gfx950 code object inside the .hip_fatbin section), finds its record loop (the backward branch
with the largest body), counts the VALU instructions of one trip (= 4 points per lane) by the price classes of DESIGN.md 5, and hashes
the kernel's instruction text.  (The trip count is static: both record-load paths - non-temporal or not - are in the body and one of
them executes, phase B runs under the in-bounds mask; the dynamic average is the PMC reading SQ_INSTS_VALU / point-waves.)

    python tools/isa_count.py [--lib path] [--write]        # --write: refresh profiles/kernel_isa.json (commit it with the kernel)

bench.py calls `kernel_facts()` at run time and compares the hash with the committed profiles/kernel_isa.json: counter readings
recorded for another kernel are marked stale on the JSON line (roofline.valu_issue.isa_matches_committed)."""
import argparse
import hashlib
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "dense-visual-odometry_amd" / "dvo_amd" / "libdvo_hip.so"
OUT = ROOT / "profiles" / "kernel_isa.json"
KERNEL = "_Z8k_reduceILi0ELi0ELb0ELb0EEv15DvoReduceParams"          # k_reduce<VARIANT 0 (V1), WMODE 0, SCALE_PASS false, SKEW false>
LLVM = Path("/opt/rocm/lib/llvm/bin")

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
# half-rate on gfx950 (measured one instruction at a time, tools/ubench/probe_orders.py -> profiles/r04_ubench_instruction_rates.txt: 1.70 ns per
# wave-instruction per SIMD at 5 waves against 1.02-1.06 for v_fma / v_mul / v_add / v_sub / v_and / v_or / v_mov_b32 / v_add_u32 / v_lshrrev /
# v_ashrrev): every conversion, bit-field op, compare, select, v_fract, LEFT shift, min / max / med3, v_perm, 64-bit move, three-operand integer op
SLOW = ("v_cvt_", "v_bfe_", "v_bfi_", "v_cmp", "v_cndmask", "v_fract_", "v_lshl", "v_floor_", "v_trunc_", "v_rndne_", "v_mad_u", "v_mul_lo", "v_mul_hi",
        "v_readfirstlane", "v_mov_b64", "v_max_", "v_min_", "v_med3_", "v_perm_", "v_and_or_", "v_or3_", "v_add3_")


def _tool(name):
    p = LLVM / name
    return str(p) if p.exists() else shutil.which(name)


def disassemble(lib: Path, kernel: str):
    """-> list of (address, mnemonic, operands) of `kernel`, or raises RuntimeError."""
    objcopy, bundler, objdump = _tool("llvm-objcopy"), _tool("clang-offload-bundler"), _tool("llvm-objdump")
    if not (objcopy and bundler and objdump):
        raise RuntimeError("llvm-objcopy / clang-offload-bundler / llvm-objdump not found")
    with tempfile.TemporaryDirectory(prefix="dvo_isa_") as tmp:
        fat, co = Path(tmp) / "fat.bin", Path(tmp) / "dev.co"
        subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + str(fat), str(lib)], check=True, capture_output=True)
        targets = subprocess.run([bundler, "--list", "--type=o", "--input=" + str(fat)], check=True, capture_output=True, text=True).stdout.split()
        gpu = [t for t in targets if "gfx950" in t]
        if not gpu:
            raise RuntimeError("no gfx950 code object in {}".format(lib))
        subprocess.run([bundler, "--unbundle", "--type=o", "--input=" + str(fat), "--targets=" + gpu[0], "--output=" + str(co)], check=True, capture_output=True)
        text = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(co)], check=True, capture_output=True, text=True).stdout
    ins, inside = [], False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            inside = (m.group(1) == kernel)
            continue
        if not inside:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if not ins:
        raise RuntimeError("kernel {} not found in {}".format(kernel, lib))
    return ins, gpu[0]


def _vgpr(tok: str):
    m = re.match(r"^-?\|?v(\d+)\|?$", tok.strip())
    return int(m.group(1)) if m else None


def bank_conflict(mn: str, ops: str) -> bool:
    """An FMA whose two MULTIPLICANDS are VGPRs of one bank (register number mod 4) - or one register, a square - issues 40-60 %
    slower on gfx950 (tools/ubench/vgpr_banks.hip, profiles/r03_ubench_vgpr_banks.txt); the accumulator's bank does not matter, and
    two-source instructions never conflict."""
    if not (mn.startswith("v_fma_f32") or mn.startswith("v_fmac_f32")):
        return False
    toks = [t for t in ops.split(",")]
    if len(toks) < 3:
        return False
    a, b = _vgpr(toks[1]), _vgpr(toks[2])
    return a is not None and b is not None and (a % 4) == (b % 4)


def classify(mn: str) -> str:
    if not mn.startswith("v_"):
        return "other"
    if mn.startswith(TRANS):
        return "trans"
    if mn.startswith(SLOW):
        return "slow"
    return "fast"


def kernel_facts(lib: Path = LIB, kernel: str = KERNEL) -> dict:
    ins, target = disassemble(lib, kernel)
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    # loops: s_cbranch_* / s_branch with a negative offset (simm16 dwords relative to the next instruction)
    # The record loop = the strongly connected component of the kernel's control-flow graph with the most VALU instructions among those
    # that stay clear of the epilogue (no LDS, no float64 instruction: those are the workgroup reduction and its little loops).  With
    # wave-uniform loop control the compiler places the four in-bounds (phase B) blocks of a trip out of line, behind the loop's own back
    # edge, each returning by a backward branch, and it parks entry-only code (the first hand-over of the record registers) between them:
    # the component holds exactly what one trip can execute.
    n = len(ins)
    leaders = {0}
    succ_of = {}
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            try:
                off = int(ops.split()[-1])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            tgt = addr_index.get(a + 4 + 4 * off)
            if tgt is not None:
                leaders.add(tgt)
                succ_of[i] = tgt
            if i + 1 < n:
                leaders.add(i + 1)
        elif mn in ("s_endpgm", "s_setpc_b64"):
            if i + 1 < n:
                leaders.add(i + 1)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for k, st in enumerate(starts):
        en = (starts[k + 1] if k + 1 < len(starts) else n) - 1
        blocks.append((st, en))
        for i in range(st, en + 1):
            block_of[i] = k
    edges = {k: set() for k in range(len(blocks))}
    for k, (st, en) in enumerate(blocks):
        mn = ins[en][1]
        if en in succ_of:
            edges[k].add(block_of[succ_of[en]])
        if mn not in ("s_branch", "s_endpgm", "s_setpc_b64") and en + 1 < n:
            edges[k].add(block_of[en + 1])
    # Tarjan, iterative
    index, low, on, stack, comps, counter = {}, {}, set(), [], [], [0]
    for root in range(len(blocks)):
        if root in index:
            continue
        work = [(root, iter(sorted(edges[root])))]
        index[root] = low[root] = counter[0]; counter[0] += 1
        stack.append(root); on.add(root)
        while work:
            v, it = work[-1]
            advanced = False
            for w in it:
                if w not in index:
                    index[w] = low[w] = counter[0]; counter[0] += 1
                    stack.append(w); on.add(w)
                    work.append((w, iter(sorted(edges[w]))))
                    advanced = True
                    break
                elif w in on:
                    low[v] = min(low[v], index[w])
            if advanced:
                continue
            work.pop()
            if work:
                low[work[-1][0]] = min(low[work[-1][0]], low[v])
            if low[v] == index[v]:
                comp = []
                while True:
                    w = stack.pop(); on.discard(w); comp.append(w)
                    if w == v:
                        break
                comps.append(comp)
    best = None
    for comp in comps:
        if len(comp) == 1 and comp[0] not in edges[comp[0]]:
            continue
        body_idx = [i for k in comp for i in range(blocks[k][0], blocks[k][1] + 1)]
        if any(ins[i][1].startswith("ds_") or "f64" in ins[i][1] for i in body_idx):
            continue
        nv = sum(1 for i in body_idx if ins[i][1].startswith("v_"))
        if best is None or nv > best[0]:
            best = (nv, sorted(body_idx))
    if best is None:
        raise RuntimeError("no loop found in {}".format(kernel))
    body = [ins[i] for i in best[1]]
    cls = {"fast": 0, "fast_bank_conflict": 0, "slow": 0, "trans": 0}
    hist = {}
    n_vmem = n_salu = n_lds = 0
    for (_, mn, ops) in body:
        c = classify(mn)
        if c == "fast" and bank_conflict(mn, ops):
            c = "fast_bank_conflict"
        if c in cls:
            cls[c] += 1
            hist[mn] = hist.get(mn, 0) + 1
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            n_vmem += 1
        elif mn.startswith("ds_"):
            n_lds += 1
        elif mn.startswith("s_"):
            n_salu += 1
    text = "\n".join("{} {}".format(mn, ops) for (_, mn, ops) in ins)
    pts = 4                                                       # DVO_PX_PER_THREAD records per lane per trip
    total = sum(cls.values())
    return {"kernel": kernel, "what": "k_reduce<V1, unweighted, zero-skew> (the level-0 launch of the metric's workload)", "target": target,
            "sha256": hashlib.sha256(text.encode()).hexdigest(), "instructions_total": len(ins),
            "loop": {"points_per_lane_per_trip": pts, "valu": total, "valu_per_point": total / pts, "by_class": cls,
                     "by_class_per_point": {k: v / pts for k, v in cls.items()}, "vmem": n_vmem, "salu": n_salu, "lds": n_lds,
                     "mnemonics": dict(sorted(hist.items(), key=lambda kv: -kv[1]))}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=str(LIB))
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    facts = kernel_facts(Path(args.lib))
    if args.write:
        import subprocess as sp
        try:
            facts["commit_of_sources"] = sp.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=str(ROOT)).stdout.strip()
        except Exception:
            pass
        OUT.write_text(json.dumps(facts, indent=1) + "\n")
    json.dump(facts, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
