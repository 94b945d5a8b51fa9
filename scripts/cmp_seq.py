"""Launch-by-launch comparison of two gpurun_out/ks/*.seq.csv files (last pass): python scripts/cmp_seq.py a b"""
import re, sys
a = [l.strip().split(',') for l in open('gpurun_out/ks/%s.seq.csv' % sys.argv[1])]
b = [l.strip().split(',') for l in open('gpurun_out/ks/%s.seq.csv' % sys.argv[2])]
npass = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = min(len(a), len(b)); per = n // npass; start = n - per
def short(x):
    m = re.search(r'(conv_\w+_kernel|gn_mish\w+|first_dgrad\w+|head\w+)', x)
    return m.group(1) if m else x[:16]
ta = tb = sa = sb = 0
for i in range(start, n):
    da, db = int(a[i][3]), int(b[i][3])
    diff = a[i][1] != b[i][1]
    sa += da; sb += db
    if diff: ta += da; tb += db
    print("%3d %-18s %8s %7.1f | %-18s %8s %7.1f %s" % (i - start, short(a[i][1]), a[i][2], da / 1e3, short(b[i][1]), b[i][2], db / 1e3, "<--" if diff else ""))
print("differing launches: %.0f vs %.0f us; all: %.0f vs %.0f us" % (ta / 1e3, tb / 1e3, sa / 1e3, sb / 1e3))
