"""Print per-kernel register / LDS / occupancy figures of libasx (hipcc remarks)."""
import re
import subprocess
import sys

src = "python-audio-separator_amd/csrc/asx.hip"
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                    "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/asx_regs.so", src],
                   capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name"):
        cur = {"name": t.split(": ")[1]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip().split(" ")[0]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("asx::", "").replace("void ", "")[:78]
    if flt and flt not in name:
        continue
    print(f"{name:78s} v{c.get('VGPRs','?'):>4} a{c.get('AGPRs','?'):>4} occ {c.get('Occupancy','?')} "
          f"sspill {c.get('SGPRs Spill','?')} vspill {c.get('VGPRs Spill','?')} scratch {c.get('ScratchSize','?')}")
