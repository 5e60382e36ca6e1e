#!/usr/bin/env python3
"""Static instruction counts of one kernel of an assembly listing, cut at marker comments (`; E16_MARK n` and the like).

    python3 profiles/isa_phase_counts.py /tmp/e16.s _Z5k_e16ILi1EE E16_MARK

Prints, per phase (the stretch between two markers, in listing order), how many vector-ALU, DPP, transcendental, LDS, memory,
scalar and wait instructions it holds.  Static counts: a loop body counts once (the labels inside a phase are listed so that the
loops can be weighted by hand)."""
import re
import sys
from collections import Counter, OrderedDict


def kind(op, line):
    if op.startswith("v_"):
        if "dpp" in op or "row_newbcast" in line or "row_shr" in line or "quad_perm" in line or "row_ror" in line:
            return "dpp"
        if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log|div_scale|div_fmas|div_fixup)", op):
            return "trans/div"
        if re.match(r"v_(fma|fmac|mul|add)_f64", op) or op.startswith("v_pk_"):
            return "f64"
        if op.startswith("v_mfma"):
            return "mfma"
        if re.match(r"v_(cmp|cndmask)", op):
            return "cmp/sel"
        if re.match(r"v_(mov|accvgpr|readlane|readfirstlane|writelane|perm|bfe|swap)", op):
            return "mov"
        return "valu-other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern, marker = sys.argv[1], sys.argv[2], sys.argv[3]
    inside = False
    phases = OrderedDict()
    cur = "start"
    phases[cur] = Counter()
    labels = {cur: []}
    for line in open(path):
        s = line.strip()
        if not inside:
            if s.startswith(kern) and re.match(r"\S+:", s):
                inside = True
            continue
        if s.startswith(".Lfunc_end"):
            break
        m = re.search(marker + r"\s+(\d+)", s)
        if m:
            cur = "mark " + m.group(1)
            phases.setdefault(cur, Counter())
            labels.setdefault(cur, [])
            continue
        if s.startswith(".LBB") and s.endswith(":"):
            labels[cur].append(s[:-1])
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        phases[cur][kind(op, s)] += 1
    cols = ["f64", "dpp", "trans/div", "cmp/sel", "mov", "valu-other", "mfma", "lds", "vmem", "salu", "branch", "wait", "nop"]
    print("%-10s %6s | " % ("phase", "VALU") + " ".join("%9s" % c for c in cols) + " | labels")
    tot = Counter()
    for ph, c in phases.items():
        valu = sum(c[k] for k in ("f64", "dpp", "trans/div", "cmp/sel", "mov", "valu-other", "mfma"))
        tot.update(c)
        print("%-10s %6d | " % (ph, valu) + " ".join("%9d" % c[k] for k in cols) + " | %d" % len(labels[ph]))
    valu = sum(tot[k] for k in ("f64", "dpp", "trans/div", "cmp/sel", "mov", "valu-other", "mfma"))
    print("%-10s %6d | " % ("total", valu) + " ".join("%9d" % tot[k] for k in cols))


if __name__ == "__main__":
    main()
