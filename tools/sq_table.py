"""tools/pmc_kernel.sh log -> one row per kernel family: share of SIMD time the MFMA pipe is busy, share of wave time spent
waiting, LDS figures.   python tools/sq_table.py gpurun_out/TAG/sq.log [steps]"""
import collections, sys
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
v = collections.defaultdict(dict)
n = {}
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 7 and p[3] == "total" and p[4] == "over":
        v[p[0]][p[1]] = float(p[2])
        n[p[0]] = int(p[5])
print("| kernel family | launches / step | kernel time / step (ms) | MFMA pipe busy | wave time in s_waitcnt | waiting for anything | of it on LDS | bank-conflict share of LDS cycles | VALU instructions per MFMA |")
print("|---|---|---|---|---|---|---|---|---|")
for k, c in sorted(v.items()):
    try:
        busy = c["SQ_BUSY_CYCLES"] / 32.0                       # summed over the 32 shader engines
        wave = c["SQ_WAVE_CYCLES"]
        print(f"| `{k}` | {n[k] / steps:.0f} | {busy / 2.1e9 / steps * 1e3:.2f} | {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (busy * 1024):.1%} | "
              f"{c['SQ_WAIT_INST_ANY'] / wave:.0%} | {c['SQ_WAIT_ANY'] / wave:.0%} | {c['SQ_WAIT_INST_LDS'] / wave:.1%} | "
              f"{c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.1%} | {c['SQ_INSTS_VALU'] / max(c['SQ_INSTS_MFMA'], 1):.1f} |")
    except KeyError as e:
        print(f"| `{k}` | missing {e} |")
