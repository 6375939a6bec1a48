"""Appends the derived figures to an SQ summary (the concatenation of the two passes of tools/pmc_sq.sh): python tools/pmc_sq_derive.py FILE
Counters are per shader-engine slice (8 CUs = 32 SIMDs on MI355X); SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles."""
import re, sys
path = sys.argv[1]
txt = open(path).read()
v = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(SQ_\w+)\s+per-dispatch\s+([0-9.]+)", txt, re.M)}
busy = v["SQ_BUSY_CYCLES"]
line = ("# derived: resident waves / SIMD = %.2f; waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES) %.0f %%; VALU busy (SQ_ACTIVE_INST_VALU x 4 / 32 SIMDs / busy) %.0f %%; "
        "LDS busy (SQ_LDS_IDX_ACTIVE / 8 CUs / busy) %.0f %%, bank conflicts %.0f %% of it; VALU instructions per wave (all waves of the grid) %.0f; "
        "whole launch: %.1f M VALU wave-instructions = %.0f us of issue at 1024 SIMDs x 2.4 GHz / 4 cycles"
        % (v["SQ_WAVE_CYCLES"] * 4 / busy / 32, 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_ACTIVE_INST_VALU"] * 4 / busy / 32,
           100 * v["SQ_LDS_IDX_ACTIVE"] / 8 / busy, 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], v["SQ_INSTS_VALU"] / v["SQ_WAVES"],
           v["SQ_INSTS_VALU"] * 32 / 1e6, v["SQ_INSTS_VALU"] * 32 / (1024 * 2.4e9 / 4) * 1e6))
if "# derived" not in txt:
    open(path, "a").write(line + "\n")
print(line)
