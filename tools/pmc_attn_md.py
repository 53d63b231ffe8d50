"""gpurun_out/pmc_attn_<tag>_{a,b}.txt (tools/pmc_attn.sh: raw per-launch counter sums) -> one markdown table per tag.
MfmaUtil % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.
usage: pmc_attn_md.py <tag> [<tag> ...]"""
import ast, re, sys
def load(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(?:void )?(\S+?)(?:<.*?>)? launches (\d+) (\{.*\})", line.strip())
        if m:
            out[m.group(1)] = {k: float(v) for k, v in ast.literal_eval(m.group(3)).items()}
    return out
for tag in sys.argv[1:]:
    a, b = load("gpurun_out/pmc_attn_%s_a.txt" % tag), load("gpurun_out/pmc_attn_%s_b.txt" % tag)
    print("### %s\n" % tag)
    print("| kernel | GPU-active cycles / launch (per XCD) | MfmaUtil % | waves parked % (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | MFMA instr / launch | VALU instr / launch | LDS instr / launch | LDS bank-conflict cycles / launch | conflict cycles per LDS-active cycle |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k in sorted(a):
        x, y = a[k], b.get(k, {})
        act = x["GRBM_GUI_ACTIVE"] / 8.0
        print("| %s | %.0f | %.1f | %.1f | %.3g | %.3g | %.3g | %.3g | %.3f |" % (
            k, act, 100.0 * x["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024.0), 100.0 * x["SQ_WAIT_ANY"] / max(x["SQ_WAVE_CYCLES"], 1.0),
            y.get("SQ_INSTS_MFMA", 0), x["SQ_INSTS_VALU"], y.get("SQ_INSTS_LDS", 0), y.get("SQ_LDS_BANK_CONFLICT", 0),
            y.get("SQ_LDS_BANK_CONFLICT", 0) / max(y.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)))
    print()
