"""Summarise -Rpass-analysis=kernel-resource-usage remarks: python tools/resusage.py LOG [substr]"""
import re, sys
t = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for b in t.split('Function Name: ')[1:]:
    name = b.split(' ')[0]
    if sub not in name:
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    m = re.search(r'kernelINS_(\w+?)E((?:Li\d+E)+)', name)
    tag = (m.group(1), re.findall(r'Li(\d+)E', m.group(2))) if m else name[:70]
    print(tag, 'vgpr', g('VGPRs'), 'agpr', g('AGPRs'), 'spill', g('VGPRs Spill'), 'scratch',
          g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'))
