"""Per-kernel-family time of two `rocprofv3 --kernel-trace --stats` runs side by side (A/B of two library builds).
python tools/probe/kstat_diff.py <dirA> <dirB>"""
import sys, glob, csv, re, collections


def load(d):
    t = collections.Counter(); c = collections.Counter()
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            k = re.sub(r"\(.*", "", n)
            if k.startswith("void conv_mma8_kernel"):
                k = re.sub(r", 8>$", ">", k)          # (the NW template parameter added in round 5)
            k = k if len(k) < 90 else k[:90]
            t[k] += float(r["TotalDurationNs"]); c[k] += int(r["Calls"])
    return t, c


ta, ca = load(sys.argv[1]); tb, cb = load(sys.argv[2])
keys = sorted(set(ta) | set(tb), key=lambda k: -(abs(tb[k] - ta[k])))
print(f"total A {sum(ta.values()) / 1e6:.2f} ms  B {sum(tb.values()) / 1e6:.2f} ms")
for k in keys[:30]:
    print(f"{(tb[k] - ta[k]) / 1e3:9.1f} us  A {ta[k] / 1e3:9.1f} ({ca[k]})  B {tb[k] / 1e3:9.1f} ({cb[k]})  {k}")
