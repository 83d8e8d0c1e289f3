"""Per-kernel sums of the SQ counters collected by tools/gpu/pmc_sq.sh (rocprofv3 --pmc, one CSV per pass) and the ratios that
say where a kernel's waves spend their cycles.   python tools/pmc_sq_summary.py <dir with p1/ p2/ ...>"""
import collections
import csv
import os
import re
import sys


def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    return k[:64]


def main(base):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for d in sorted(os.listdir(base)):
        p = os.path.join(base, d)
        if not os.path.isdir(p):
            continue
        for root, _, files in os.walk(p):
            for f in files:
                if not f.endswith("counter_collection.csv"):
                    continue
                seen = set()
                for r in csv.DictReader(open(os.path.join(root, f))):
                    k = short(r["Kernel_Name"])
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    key = (k, r.get("Dispatch_Id"))
                    if d == "p1" and key not in seen:
                        seen.add(key)
                        cnt[k] += 1
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0))[:14]
    print("kernel, launches | per WAVE-cycle: waiting on any instruction / on LDS, issuing VALU / LDS / VMEM | MFMA pipe busy % of GPU time | "
          "instructions per MFMA: VALU (incl. MFMA), LDS, VMEM reads, SALU | LDS bank-conflict cycles per LDS-active cycle | "
          "VMEM reads in flight per wave (INST_LEVEL_VMEM / WAVE_CYCLES) | L2 (TCC) hit rate, requests, reads sent to the fabric | L1 (TCP) miss "
          "rate and mean L1->L2 read latency")
    for k, c in rows:
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        mf = c.get("SQ_INSTS_MFMA", 0.0) or 1.0
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) or 1.0
        busy = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024)   # per-SIMD busy cycles summed over 1024 SIMDs; GUI_ACTIVE over 8 XCDs
        print(f"{k:64s} n={cnt[k]:4d} | wait_any {c.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} wait_lds {c.get('SQ_WAIT_INST_LDS', 0) / wc:5.2f} "
              f"valu {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.2f} lds {c.get('SQ_ACTIVE_INST_LDS', 0) / wc:5.2f} "
              f"vmem {c.get('SQ_ACTIVE_INST_VMEM', 0) / wc:5.2f} | mfma_busy {busy:5.1f} % | per MFMA: valu {c.get('SQ_INSTS_VALU', 0) / mf:6.1f} "
              f"lds {c.get('SQ_INSTS_LDS', 0) / mf:5.2f} vmem_rd {c.get('SQ_INSTS_VMEM_RD', 0) / mf:5.2f} salu {c.get('SQ_INSTS_SALU', 0) / mf:6.1f} | "
              f"bank_conf {c.get('SQ_LDS_BANK_CONFLICT', 0) / (c.get('SQ_LDS_IDX_ACTIVE', 0) or 1):5.2f} | "
              f"vmem_level {c.get('SQ_INST_LEVEL_VMEM', 0) / wc:5.2f} | L2 hit {100.0 * c.get('TCC_HIT', 0) / ((c.get('TCC_HIT', 0) + c.get('TCC_MISS', 0)) or 1):5.1f} % "
              f"L2 req/launch {c.get('TCC_REQ', 0) / max(cnt[k], 1) / 1e6:7.2f} M, to fabric {c.get('TCC_EA0_RDREQ', 0) / max(cnt[k], 1) / 1e6:6.2f} M | "
              f"L1 miss {100.0 * c.get('TCP_TCC_READ_REQ', 0) / (c.get('TCP_TOTAL_CACHE_ACCESSES', 0) or 1):5.1f} % "
              f"L1->L2 read latency {c.get('TCP_TCC_READ_REQ_LATENCY', 0) / (c.get('TCP_TCC_READ_REQ', 0) or 1):6.0f} cycles")


if __name__ == "__main__":
    main(sys.argv[1])
