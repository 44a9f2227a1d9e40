"""Per-kernel SQ stall breakdown from a rocprofv3 --pmc pass (profiles/r01_pmc_sq.json).

Counters (one pass, 8 SQ slots; no tracing flags): SQ_WAVE_CYCLES, SQ_WAIT_ANY (wave parked on s_waitcnt / barrier),
SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY (issuing), SQ_INSTS_VALU, SQ_INSTS_VMEM_RD, SQ_WAVES, SQ_BUSY_CYCLES.
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (guide, rocprofv3 PMC slots section).
usage: pmc_sq.py <counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

from pmc_traffic import short

NAMES = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD",
         "SQ_WAVES", "SQ_BUSY_CYCLES")


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] in NAMES:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, d in sorted(acc.items()):
        v = {n: sum(x) / len(x) for n, x in d.items()}
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        out[k] = dict(v, frac_parked_on_waitcnt=v.get("SQ_WAIT_ANY", 0.0) / wc,
                      frac_issue_stalled=v.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                      frac_issuing=v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                      valu_per_wave=v.get("SQ_INSTS_VALU", 0.0) / (v.get("SQ_WAVES", 0.0) or 1.0))
        print(f"{k:16s} parked {out[k]['frac_parked_on_waitcnt']:.2f}  issue-stalled {out[k]['frac_issue_stalled']:.2f}  "
              f"issuing {out[k]['frac_issuing']:.2f}  VALU/wave {out[k]['valu_per_wave']:.0f}  waves {v.get('SQ_WAVES', 0):.0f}")
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
