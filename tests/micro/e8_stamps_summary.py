"""Per-phase cycle summary of encoder8_kernel from a -DENC8_STAMPS run of tests/micro/enc_stamps.py (device printf)."""
import statistics, sys
v = [int(l.split()[2]) for l in open(sys.argv[1]) if l.startswith("e8 ")]
r = v[-85:]
names = ["P0 norm+concat", "P1 qkv", "P2 attention", "P3 o-proj", "P4 norm", "P5 ffn1", "P6 ffn2"]
print("prologue", r[0])
tot = 0
for k in range(7):
    xs = [r[1 + 7 * l + k] for l in range(12)]
    m = statistics.median(xs); tot += m
    print(f"{names[k]:16s} median {m:7.0f}  min {min(xs)} max {max(xs)}")
print("layer median sum", tot, " whole", sum(r))
