#!/usr/bin/env python
"""Static ISA census of the kernels bench.py times: instruction counts by issue class inside the
time-step loop of each dispatched instantiation, from `llvm-objdump -d` of the gfx950 code object inside the built csrc/*.o, priced
at the issue rates of /opt/skills/guides/MI355X_MICROARCH.md -> the per-launch time the VALU pipe and the matrix pipe need AT LEAST
for this instruction stream (`valu_ceiling_ms`, `mfma_ceiling_ms` of bench.py's roofline records).

    python tools/isa_census.py            # -> profiles/r06_isa_census.json (+ a table on stdout)
    python tools/isa_census.py --list ncde_fast      # kernel symbols of one object

How a count becomes a time: natural loops of the kernel's control-flow graph (dominator analysis on the disassembly); every
instruction is weighted by the product of the trip counts of the loops around it; trip counts come from the workload (`table()` below,
in program order of the loop heads; loops not listed -- spin loops on LDS flags, prologue fills -- count once).  Issue cost per wave64 instruction on a SIMD-32: plain VALU 2 cycles (4 for the packed-fp32 `v_pk_*_f32` forms:
no rate gain on CDNA3/4), transcendentals (v_exp / v_rcp / v_rsq / v_sqrt / v_log / v_sin / v_cos) 8 (quarter rate),
v_mfma_*_16x16x32_{f16,bf16} 16, 32x32x16 32, f32-input 16x16x4 32, 32x32x2 64 (matrix pipe, per SIMD).  The SIMD serves
`waves_per_simd` waves of the workgroup, each issuing the same stream, and the chip runs ceil(workgroups / (256 CUs x wgs per CU))
rounds: ceiling = rounds x waves_per_simd x cycles / 2.4 GHz.  It is a floor on the kernel time for THIS instruction stream (no stall,
perfect dual issue of the two pipes), which is what makes achieved / ceiling a roofline fraction <= 1.
`serial_issue_ceiling_ms` (round 6): measured on MI355X (tools/mfma_valu_overlap.hip, profiles/r06_mfma_valu_overlap.txt), the matrix
pipe and the VALU of a SIMD do NOT work side by side: next to a saturated MFMA stream the other wave of the SIMD issues one VALU
instruction per 12.8 cycles (4.3 alone), two mixed streams on one SIMD finish one AFTER the other (the older wave has priority), and
inside one wave an independent FMA next to every MFMA costs 4.1 cycles on top of the MFMA's 17.  So the floor of a mixed stream is the
SUM over the waves of a SIMD of: VALU 4.1, packed fp32 6.5, transcendental 10.0, MFMA 16x16x32 17.1, 32x32x16 / 16x16x4-f32 32.1
cycles per instruction -- not the maximum of the two pipes.
`issue_ceiling_ms`: a wave issues at most one instruction per 4-cycle slot (MI355X_MICROARCH.md: "32 cyc/SIMD ~ 8 issue slots of ~4
cyc"), so its own stream -- every class, waits and nops included -- takes instructions x 4 cycles at least; for kernels whose waves
play different roles the longest role counts.  With ONE wave per SIMD (the register-resident kernels) this, not a pipe, is the floor.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"
CLOCK_HZ = 2.4e9
N_CU = 256

# measured serial issue cost per wave64 instruction when MFMA and VALU work share a SIMD (tools/mfma_valu_overlap.hip)
SERIAL = {"valu": 4.1, "valu_pk": 6.5, "trans": 10.0, "mfma": {16: 17.1, 32: 32.1, 64: 64.0}}

TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")


def classify(mn):
    """-> (class, SIMD cycles per wave64 instruction)"""
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        if "32x32x2" in mn and "f32" in mn.split("32x32x2")[1][:5]:
            return "mfma", 64
        if "16x16x4" in mn and mn.endswith("f32"):
            return "mfma", 32
        if "32x32x" in mn:
            return "mfma", 32
        return "mfma", 16
    if mn.startswith("v_"):
        if mn.startswith(TRANS):
            return "trans", 8
        if mn.startswith("v_pk_") and mn.endswith("_f32"):
            return "valu_pk", 4
        if mn.startswith("v_accvgpr") or mn.startswith("v_nop"):
            return "valu", 2
        return "valu", 2
    if mn.startswith("ds_"):
        return "lds", 0
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0
    if mn.startswith("s_waitcnt") or mn.startswith("s_nop") or mn.startswith("s_barrier"):
        return "sync", 0
    if mn.startswith("s_"):
        return "salu", 0
    return "other", 0


def code_object(obj):
    """The gfx950 code object bundled in a host object file -> path of a temporary copy."""
    tmp = tempfile.mkdtemp(prefix="census_")
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        if f.endswith("gfx950"):
            return os.path.join(tmp, f)
    raise RuntimeError("no gfx950 bundle in " + obj)


_DIS = {}


def disassemble(obj):
    """{symbol: [(addr, mnemonic, operand text)]} of every kernel in the object's device code."""
    if obj in _DIS:
        return _DIS[obj]
    co = code_object(obj)
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    shutil.rmtree(os.path.dirname(co), ignore_errors=True)
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    _DIS[obj] = out
    return out


def _target(a, ops):
    m = re.match(r"^(-?\d+)", ops.strip())
    if not m:
        return None
    simm = int(m.group(1))
    if simm >= 32768:
        simm -= 65536
    return a + 4 + 4 * simm


def loops_of(ins):
    """NATURAL loops of the kernel's control-flow graph: [(head index, sorted member instruction indices)], in program order of
    the heads.  (Backward branches alone will not do: hipcc places cold blocks behind the code that jumps to them and back.)
    Basic blocks from branch targets / fall-throughs; dominators by the iterative algorithm; a back edge is an edge whose target
    dominates its source; the loop of a back edge = every block that reaches its source without passing the head; loops that share a
    head are merged."""
    n = len(ins)
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    leaders = {0}
    succ_of = {}
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = _target(a, ops)
            ti = idx.get(t)
            nxt = [i + 1] if (mn != "s_branch" and i + 1 < n) else []
            succ_of[i] = nxt + ([ti] if ti is not None else [])
            if ti is not None:
                leaders.add(ti)
            if i + 1 < n:
                leaders.add(i + 1)
        elif mn == "s_endpgm":
            succ_of[i] = []
            if i + 1 < n:
                leaders.add(i + 1)
    starts = sorted(leaders)
    bid = {}
    blocks = []
    for k, st in enumerate(starts):
        en = (starts[k + 1] if k + 1 < len(starts) else n) - 1
        blocks.append((st, en))
        for i in range(st, en + 1):
            bid[i] = k
    nb = len(blocks)
    succ = [[] for _ in range(nb)]
    for k, (st, en) in enumerate(blocks):
        outs = succ_of.get(en)
        if outs is None:
            outs = [en + 1] if en + 1 < n else []
        succ[k] = sorted(set(bid[o] for o in outs))
    pred = [[] for _ in range(nb)]
    for k in range(nb):
        for t in succ[k]:
            pred[t].append(k)
    # reachable blocks, reverse post-order
    seen, order = set(), []
    stack = [(0, iter(succ[0]))]
    seen.add(0)
    while stack:
        k, it = stack[-1]
        for t in it:
            if t not in seen:
                seen.add(t)
                stack.append((t, iter(succ[t])))
                break
        else:
            order.append(k)
            stack.pop()
    rpo = order[::-1]
    pos = {k: i for i, k in enumerate(rpo)}
    idom = {0: 0}
    changed = True
    while changed:      # Cooper / Harvey / Kennedy
        changed = False
        for k in rpo[1:]:
            ps = [q for q in pred[k] if q in idom]
            if not ps:
                continue
            new = ps[0]
            for q in ps[1:]:
                a_, b_ = new, q
                while a_ != b_:
                    while pos[a_] > pos[b_]:
                        a_ = idom[a_]
                    while pos[b_] > pos[a_]:
                        b_ = idom[b_]
                new = a_
            if idom.get(k) != new:
                idom[k] = new
                changed = True

    def dominates(h, k):
        while True:
            if k == h:
                return True
            if k == 0 or k not in idom:
                return False
            k = idom[k]

    by_head = {}
    for k in rpo:
        for t in succ[k]:
            if t in idom and dominates(t, k):      # back edge k -> t
                body = by_head.setdefault(t, {t})
                work = [k]
                while work:
                    q = work.pop()
                    if q in body:
                        continue
                    body.add(q)
                    work.extend(p_ for p_ in pred[q] if p_ in seen)
    loops = []
    for h in sorted(by_head, key=lambda b: blocks[b][0]):
        members = sorted(i for b in by_head[h] for i in range(blocks[b][0], blocks[b][1] + 1))
        loops.append((blocks[h][0], members))
    return loops


def census(ins, trips):
    """Weighted class counts.  `trips`: trip count per natural loop in program order of the loop heads (nesting = containment), missing
    entries count once."""
    loops = loops_of(ins)
    weight = [1.0] * len(ins)
    used = []
    for k, (h, members) in enumerate(loops):
        n = trips[k] if k < len(trips) else 1
        used.append({"head": "%x" % ins[h][0], "instructions": len(members), "trip": n})
        for i in members:
            weight[i] *= n
    counts, cycles = Counter(), Counter()
    for (a, mn, ops), w in zip(ins, weight):
        cls, cyc = classify(mn)
        counts[cls] += w
        if cls == "mfma":
            cycles["mfma"] += w * cyc
            cycles["serial"] += w * SERIAL["mfma"][cyc]
        elif cyc:
            cycles["valu"] += w * cyc
            cycles["serial"] += w * SERIAL[cls]
    return counts, cycles, used


def find_symbol(dis, fragment):
    hits = [s for s in dis if fragment in s]
    if len(hits) != 1:
        raise RuntimeError("%d symbols match %r: %s" % (len(hits), fragment, hits[:4]))
    return hits[0]


# record key -> (object, mangled-name fragment, waves per SIMD, workgroups of the launch, workgroups a CU holds, trip counts of the
# natural loops in program order of their heads, roles = groups of loop indices that only one kind of wave executes).
# Shapes: BASELINE cfg2 / cfg4 as bench.py runs them (the batch-tiled kernels of cfg5 have data-dependent loop nests: their
# instruction counts come from the SQ_INSTS_* counters instead, tools/pmc_summary.py).
def table():
    T2, T4 = 399, 182
    return {
        # cfg2: B = 4096 -> 256 workgroups of 16 samples.  Forward: ONE loop = the 398 steps, its body = the 4 stages unrolled.
        "cfg2.forward": dict(obj="ncde_fast_fwd3", frag="ncde_fwd_fast_bf3ILi32ELi32ELi20ELi4ELi0ELi2ELi0ELi3ELi1ELi0E", waves_per_simd=1, n_wg=256, wg_per_cu=1,
                             trips=[T2 - 1]),
        # adjoint: loop 0 prologue fill; loops 1-2 = the gradient waves' step / stage loops; 3-8 their flag polls; 9-10 = the chain
        # waves' step / stage loops.  One wave of each role per SIMD.
        "cfg2.backward": dict(obj="ncde_fast", frag="ncde_adj_fast3ILi3ELi20ELi0ELi2ELi0ELi0ELi2ELi0E", waves_per_simd=2, n_wg=256, wg_per_cu=1,
                              trips=[1, T2 - 1, 4, 1, 1, 1, 1, 1, 1, T2 - 1, 4], roles=[[1, 2, 3, 4, 5, 6, 7, 8], [9, 10]]),
        # cfg4: B = 8192 -> 512 workgroups; forward 2 per CU (4 waves each), backward 1 per CU (two rounds); midpoint: 2 stages
        "cfg4.forward": dict(obj="ncde_fast_fwd3", frag="ncde_fwd_fast_bf3ILi64ELi64ELi4ELi4ELi1ELi1ELi0ELi3ELi1ELi0E", waves_per_simd=1, n_wg=512, wg_per_cu=2,
                             trips=[T4 - 1]),
        "cfg4.backward": dict(obj="ncde_fast64", frag="ncde_adj_h64ILi1ELi1ELi1ELi0ELi0E", waves_per_simd=1, n_wg=512, wg_per_cu=1, trips=[T4 - 1, 2]),
    }


def run(entry):
    dis = disassemble(os.path.join(ROOT, "online-neural-cdes_amd", "csrc", entry["obj"] + ".o"))
    sym = find_symbol(dis, entry["frag"])
    ins = dis[sym]
    counts, cycles, loops = census(ins, entry["trips"])
    if len(loops) != len(entry["trips"]):
        raise RuntimeError("%s: %d natural loops, the table lists %d trip counts -- the kernel changed, update table()" % (sym, len(loops), len(entry["trips"])))
    rounds = -(-entry["n_wg"] // (N_CU * entry["wg_per_cu"]))
    roles = entry.get("roles")
    # VALU / matrix pipe of a SIMD: shared by wg_per_cu x waves_per_simd waves.  A kernel whose waves play different ROLES (chain /
    # gradient waves of ncde_adj_fast3) holds every role's code in one stream behind a wave-uniform branch; a SIMD hosts one wave
    # of each role, so the weighted stream as a whole is what ONE SIMD executes per workgroup.
    share = entry["wg_per_cu"] * entry["waves_per_simd"] / float(len(roles) if roles else 1)
    # issue: per wave, 4 cycles per instruction of any class; the longest role
    lp = loops_of(ins)
    weight = [1.0] * len(ins)
    for k, (h, members) in enumerate(lp):
        for i in members:
            weight[i] *= entry["trips"][k]
    if roles:
        per_role = []
        for grp in roles:
            member_set = set(i for k in grp for i in lp[k][1])
            per_role.append(sum(weight[i] for i in member_set))
        issue_ins = max(per_role)
    else:
        issue_ins = sum(weight)
    res = {"symbol": sym, "instructions_static": len(ins), "loops": loops,
           "per_launch_weighted": {k: round(v, 1) for k, v in counts.items()},
           "valu_cycles_per_simd_and_workgroup": round(cycles["valu"], 1), "mfma_cycles_per_simd_and_workgroup": round(cycles["mfma"], 1),
           "waves_sharing_a_simd": share, "rounds": rounds, "instructions_issued_by_the_longest_wave": round(issue_ins, 1),
           "valu_ceiling_ms": round(rounds * share * cycles["valu"] / CLOCK_HZ * 1e3, 4),
           "mfma_ceiling_ms": round(rounds * share * cycles["mfma"] / CLOCK_HZ * 1e3, 4),
           "issue_ceiling_ms": round(rounds * issue_ins * 4 / CLOCK_HZ * 1e3, 4),
           "serial_issue_ceiling_ms": round(rounds * share * cycles["serial"] / CLOCK_HZ * 1e3, 4)}
    return res


def census_for(keys=None):
    """{key: record} for the table's entries (bench.py calls this live on the GPU box: the objects travel with the snapshot)."""
    return {k: run(e) for k, e in table().items() if keys is None or k in keys}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", default=None, help="object name (e.g. ncde_fast): print its kernel symbols and loop structure")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_isa_census.json"))
    a = ap.parse_args()
    if a.list:
        dis = disassemble(os.path.join(ROOT, "online-neural-cdes_amd", "csrc", a.list + ".o"))
        for s, ins in dis.items():
            lp = loops_of(ins)
            print("%6d ins  %2d loops  %s" % (len(ins), len(lp), s))
            for h, members in lp:
                cnt = Counter(classify(ins[i][1])[0] for i in members)
                print("          loop head %x: %d ins (%s)" % (ins[h][0], len(members), ", ".join("%s %d" % kv for kv in sorted(cnt.items()))))
        return
    from ncde_amd import _lib
    out = {"_meta": {"source_fingerprint": _lib.source_fingerprint(), "clock_hz": CLOCK_HZ,
                     "issue_cycles": {"valu": 2, "valu_pk_f32": 4, "transcendental": 8, "mfma_16x16x32_16bit": 16, "mfma_32x32x16_16bit": 32,
                                      "mfma_16x16x4_f32": 32, "mfma_32x32x2_f32": 64},
                     "serial_issue_cycles": {"valu": 4.1, "valu_pk_f32": 6.5, "transcendental": 10.0, "mfma_16x16x32_16bit": 17.1, "mfma_32x32x16_16bit": 32.1,
                                             "mfma_16x16x4_f32": 32.1, "source": "profiles/r06_mfma_valu_overlap.txt (tools/mfma_valu_overlap.hip)"},
                     "note": "static census by tools/isa_census.py; ceilings are floors on the kernel time for the instruction stream as compiled"}}
    for name, entry in table().items():
        out[name] = run(entry)
        r = out[name]
        print("%-14s valu %.3f ms  mfma %.3f ms  issue %.3f ms  serial issue (MFMA + VALU of a SIMD add up) %.3f ms  (weighted per launch: valu %d trans %d pk %d mfma %d)" % (
            name, r["valu_ceiling_ms"], r["mfma_ceiling_ms"], r["issue_ceiling_ms"], r["serial_issue_ceiling_ms"], r["per_launch_weighted"].get("valu", 0),
            r["per_launch_weighted"].get("trans", 0), r["per_launch_weighted"].get("valu_pk", 0), r["per_launch_weighted"].get("mfma", 0)))
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
