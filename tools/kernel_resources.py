"""Developer tool: VGPR / LDS / occupancy of selected kernel instantiations from hipcc's -Rpass-analysis=kernel-resource-usage
output (stdin or a file): python tools/kernel_resources.py /tmp/attn_res.txt 'Li32EDF16bDF16bLi4ELb1E'"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    if pat in name:
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, None])[1]
        print(name[:60], "VGPR", g("VGPRs"), "SGPR", g("SGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"), "occ", g(r"Occupancy \[waves/SIMD\]"),
              "LDS", g(r"LDS Size \[bytes/block\]"))
