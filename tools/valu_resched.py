#!/usr/bin/env python3
"""Post-register-allocation rescheduler for the gfx950 assembly of the kernels (an experiment: `make -C dense-visual-odometry_amd/csrc resched OUT=...`; measured and NOT adopted, profiles/r04_experiments.txt).

Why.  On gfx950 a simple float32 VALU instruction issues in ~1.05 ns per wave per SIMD, but three classes cost ~1.7 ns when they
follow one another: conversions / bit-field extracts / compares / selects / shifts, FMAs whose two multiplicands share a VGPR bank,
and transcendental ops - while the same instructions cost (almost) nothing extra when a simple instruction sits between them
(tools/ubench/probe_orders.py, profiles/r04_ubench_probe_orders.txt: the level-0 record loop's own composition issues in 736 ns per
trip with the heavy instructions clustered, 637 ns with them spread).  hipcc's scheduler knows no such classes: it emits the texel
unpack of a point as one cluster of ~20 conversions, and the bank-conflicted accumulations in runs.

What.  For every maximal run of plain VALU instructions inside a basic block (no branch, label, wait, scalar, memory, LDS, DPP /
SDWA / lane-crossing, transcendental or inline-asm instruction in it - those end a run and stay where they are), a dependency-
respecting list scheduling over the run's own instructions that avoids placing a heavy instruction directly behind another heavy one
whenever a ready simple instruction exists, and otherwise keeps the compiler's order.  Registers are physical (post-RA), so the
dependencies are exact: read-after-write, write-after-read and write-after-write on every VGPR, SGPR, VCC, EXEC, SCC and M0 an
instruction names or implies.  Nothing is added, removed or changed: the output is a permutation of each run.

    valu_resched.py in.s out.s [--report]
"""
import re
import sys

HEAVY_PREFIX = ("v_cvt_", "v_bfe_", "v_cmp", "v_cndmask", "v_fract_", "v_lshl", "v_lshr", "v_ashr", "v_floor_", "v_trunc_", "v_rndne_", "v_mad_u",
                "v_mul_lo", "v_mul_hi", "v_mov_", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_min_f64")
# never part of a run: lane-crossing, modifiers that read neighbours, ops with extra hazards, transcendental, accumulation registers
EXCLUDE_SUBSTR = ("_dpp", "_sdwa", "readlane", "readfirstlane", "writelane", "permlane", "v_cmpx", "v_div_", "v_nop", "v_mfma", "v_accvgpr", "v_swap",
                  "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_mbcnt", "v_bcnt", "v_movrel", "v_interp", "v_dot", "v_pk_", "v_smfmac")

_REG_RE = re.compile(r"(?<![A-Za-z0-9_.])(v|s|a)(\d+)(?![A-Za-z0-9_])|(?<![A-Za-z0-9_.])(v|s|a)\[(\d+):(\d+)\]|\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|scc|m0)\b")


def regs_of(text):
    out = set()
    for m in _REG_RE.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        elif m.group(3):
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add(m.group(3) + str(k))
        else:
            r = m.group(6)
            out.add({"vcc_lo": "vcc", "vcc_hi": "vcc", "exec_lo": "exec", "exec_hi": "exec"}.get(r, r))
    return out


def parse(line):
    """-> (mnemonic, operand list) of an instruction line, or None for labels / directives / comments / blanks."""
    code = line.split("//")[0].split(";")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None
    parts = code.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return mn, ops


def is_plain_valu(mn):
    return mn.startswith("v_") and not any(x in mn for x in EXCLUDE_SUBSTR)


def n_dest(mn):
    # VOP3 forms with a second (scalar) destination
    if mn.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32", "v_mad_i64_i32")) and mn.endswith("_e64"):
        return 2
    return 1


def defs_uses(mn, ops):
    """Registers written / read (explicit operands + the implicit ones of the VOP encodings)."""
    d, u = set(), set()
    nd = n_dest(mn)
    if mn.startswith("v_cmp"):
        if mn.endswith("_e32"):
            d.add("vcc")
            for o in ops[1:] if (ops and ops[0].startswith("vcc")) else ops:
                u |= regs_of(o)
        else:
            d |= regs_of(ops[0])
            for o in ops[1:]:
                u |= regs_of(o)
    else:
        for k, o in enumerate(ops):
            (d if k < nd else u).update(regs_of(o))
        if mn.startswith(("v_fmac_", "v_mac_", "v_fmamk", "v_pk_fmac")) or mn.startswith("v_fmac"):
            u |= regs_of(ops[0])                       # the destination is the accumulator too
        if mn.startswith("v_cndmask") and mn.endswith("_e32"):
            u.add("vcc")
        if mn.startswith(("v_addc_", "v_subb_", "v_subbrev_")) and mn.endswith("_e32"):
            u.add("vcc")
            d.add("vcc")
        if mn.startswith(("v_add_co", "v_sub_co", "v_subrev_co")) and mn.endswith("_e32"):
            d.add("vcc")
    u.add("exec")                                      # every VALU instruction runs under EXEC
    return d, u


def bank_conflict(mn, ops):
    if not (mn.startswith("v_fma_f32") or mn.startswith("v_fmac_f32")) or len(ops) < 3:
        return False
    def vg(t):
        m = re.match(r"^-?\|?v(\d+)\|?$", t.strip())
        return int(m.group(1)) if m else None
    a, b = vg(ops[1]), vg(ops[2])
    return a is not None and b is not None and (a % 4) == (b % 4)


def is_heavy(mn, ops):
    return mn.startswith(HEAVY_PREFIX) or bank_conflict(mn, ops)


def schedule_run(run, lookahead=48):
    """run: list of (line, mn, ops).  Returns a permutation index list."""
    n = len(run)
    if n < 3:
        return list(range(n))
    du = [defs_uses(mn, ops) for (_, mn, ops) in run]
    heavy = [is_heavy(mn, ops) for (_, mn, ops) in run]
    if not any(heavy):
        return list(range(n))
    preds = [set() for _ in range(n)]
    last_def, last_uses = {}, {}
    for i, (d, u) in enumerate(du):
        for r in u:
            if r in last_def:
                preds[i].add(last_def[r])                   # read after write
        for r in d:
            if r in last_def:
                preds[i].add(last_def[r])                   # write after write
            for j in last_uses.get(r, ()):
                if j != i:
                    preds[i].add(j)                         # write after read
        for r in u:
            last_uses.setdefault(r, []).append(i)
        for r in d:
            last_def[r] = i
            last_uses[r] = []
    remaining = [len(p) for p in preds]
    succs = [[] for _ in range(n)]
    for i, p in enumerate(preds):
        for j in p:
            succs[j].append(i)
    done = [False] * n
    order = []
    prev_heavy = False
    first = 0                                                # lowest unscheduled index
    while len(order) < n:
        while done[first]:
            first += 1
        pick = None
        if remaining[first] == 0 and not (prev_heavy and heavy[first]):
            pick = first
        else:
            # the instruction next in the compiler's order would put a heavy one behind a heavy one (or is not ready): the earliest
            # ready instruction that does not, within the lookahead; failing that, the earliest ready one
            fallback = None
            for k in range(first, min(n, first + lookahead)):
                if done[k] or remaining[k] != 0:
                    continue
                if fallback is None:
                    fallback = k
                if not (prev_heavy and heavy[k]):
                    pick = k
                    break
            if pick is None:
                pick = fallback
        done[pick] = True
        order.append(pick)
        prev_heavy = heavy[pick]
        for s in succs[pick]:
            remaining[s] -= 1
    return order


def process(lines):
    out, run, prefix = [], [], []
    stats = {"runs": 0, "instructions_in_runs": 0, "moved": 0, "heavy_pairs_before": 0, "heavy_pairs_after": 0}
    in_asm = False

    def flush():
        nonlocal run
        if run:
            order = schedule_run(run)
            h = [is_heavy(mn, ops) for (_, mn, ops) in run]
            stats["runs"] += 1
            stats["instructions_in_runs"] += len(run)
            stats["moved"] += sum(1 for a, b in zip(order, range(len(run))) if a != b)
            stats["heavy_pairs_before"] += sum(1 for a, b in zip(h, h[1:]) if a and b)
            hn = [h[i] for i in order]
            stats["heavy_pairs_after"] += sum(1 for a, b in zip(hn, hn[1:]) if a and b)
            out.extend(run[i][0] for i in order)
            run = []

    for line in lines:
        if "#ASMSTART" in line:
            in_asm = True
        p = None if in_asm else parse(line)
        if "#ASMEND" in line:
            in_asm = False
            flush()
            out.append(line)
            continue
        if p is not None and is_plain_valu(p[0]):
            run.append(("\n".join(prefix + [line]), p[0], p[1]))
            prefix = []
        else:
            # comment-only lines inside a run travel with the instruction that follows them; everything else ends the run
            if p is None and run and not in_asm and line.strip().startswith((";", "//")) and "#ASM" not in line:
                prefix.append(line)
                continue
            flush()
            out.extend(prefix)
            prefix = []
            out.append(line)
    flush()
    out.extend(prefix)
    return out, stats


def main():
    src, dst = sys.argv[1], sys.argv[2]
    with open(src) as f:
        lines = f.read().split("\n")
    out, stats = process(lines)
    with open(dst, "w") as f:
        f.write("\n".join(out))
    if "--report" in sys.argv:
        print("valu_resched: %(runs)d runs, %(instructions_in_runs)d instructions, %(moved)d moved; heavy-behind-heavy pairs %(heavy_pairs_before)d -> %(heavy_pairs_after)d" % stats)


if __name__ == "__main__":
    main()
