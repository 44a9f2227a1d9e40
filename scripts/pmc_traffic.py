"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM byte counts (profiles/rNN_pmc_traffic.json).

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 B per count?  -> they are reported in kilobytes; collected in SEPARATE passes (TCC slots: FETCH 3, WRITE 2);
on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x (TCC_EA0_RDREQ x 64 B for 128-B
requests) and must be doubled; WRITE_SIZE is exact for streaming stores.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [steps]
The bit-exact state chain (k_sb_async: one launch per step; k_sb_sys + k_sb_delta launches when the barrier-free form is
switched off) is one unit of work: its entry "fwd_state" is the SUM over its launches of a step (total / steps), like
bench.py prices it.
"""
import collections
import csv
import json
import sys

SHORT = {"k_stats": "stats", "k_resid": "residuals", "k_export_tiled": "export_natural", "FwdTrendFused": "fwd_chain", "FwdPTrend": "fwd_cov_chain",
         "FwdXTrend": "fwd_state_chain", "BwdTrend": "bwd_chain", "k_fwd_dstat": "fwd_dstat", "k_bwd_lag": "bwd_lagcov", "k_fill_rows": "pnoise_fill",
         "k_copy_active": "ecm_commit_kappa", "k_sb_": "fwd_state", "k_import_tiled": "state_reblock_out"}
PER_STEP = {"fwd_state"}


def short(name):
    if "k_chain_fix" in name:
        return None
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def collect(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        s = short(r["Kernel_Name"])
        if s:
            acc[s].append(float(r["Counter_Value"]))
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    per_launch = {k: (sum(v) / steps if k in PER_STEP else sum(v) / len(v)) for k, v in acc.items()}
    per_step = {k: sum(v) / steps for k, v in acc.items()}       # a pipelined step launches its tail kernels once per group of chains
    return per_launch, per_step


def main():
    fetch, fetch_step = collect(sys.argv[1], "FETCH_SIZE")
    write, write_step = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_kb, w_kb = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {
            "FETCH_SIZE_kb_raw": f_kb, "WRITE_SIZE_kb_raw": w_kb,
            "fetch_bytes_corrected": 2.0 * f_kb * 1024.0,      # gfx950: x2 (guide, HBM section)
            "write_bytes": w_kb * 1024.0,
            "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
            "hbm_bytes_per_step": 2.0 * fetch_step.get(k, 0.0) * 1024.0 + write_step.get(k, 0.0) * 1024.0,
        }
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        print(f"{k:18s} fetch(raw) {v['FETCH_SIZE_kb_raw']/1e6:8.3f} GB-ish  write {v['write_bytes']/1e9:8.3f} GB  "
              f"hbm(corrected) {v['hbm_bytes_per_launch']/1e9:8.3f} GB per launch, {v['hbm_bytes_per_step']/1e9:8.3f} GB per step")
    print(f"{'sum per step':18s} {sum(v['hbm_bytes_per_step'] for v in out.values())/1e9:8.3f} GB")


if __name__ == "__main__":
    main()
