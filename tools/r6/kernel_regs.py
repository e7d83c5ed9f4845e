"""Per-kernel register / spill / LDS table of a hipcc -S --cuda-device-only assembly file (the amdhsa.kernels metadata)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
md = txt[txt.index("amdhsa.kernels:"):]
names, rows = [], []
for k in md.split("  - .agpr_count:")[1:]:
    g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", k).group(1))
    names.append(re.search(r"\.name:\s+(\S+)", k).group(1))
    rows.append((g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count")))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
match = sys.argv[2] if len(sys.argv) > 2 else ""
for d, (v, sp, ss) in zip(dem, rows):
    if match in d:
        print(f"{v:4d} vgpr {sp:4d} spilled {ss:4d} sgpr-spilled  {d[:150]}")
