"""Per-phase cycle summary of encoder8_kernel from a -DENC8_STAMPS run of tests/micro/enc_stamps.py (device printf).
Stamps (round 3): prologue, then per layer: behind the barrier at the head of P0 (= P6 of the layer before; embedding for
layer 0), behind P0, P1, P2, P3, P4, P5; one more behind the last head barrier."""
import statistics, sys
v = [int(l.split()[2]) for l in open(sys.argv[1]) if l.startswith("e8 ")]
r = v[-86:]
names = ["P6 ffn2 (prev)", "P0 norm+concat", "P1 qkv", "P2 attention", "P3 o-proj", "P4 norm", "P5 ffn1"]
print("prologue", r[0])
tot = 0
for k in range(7):
    xs = [r[1 + 7 * l + k] for l in range(1 if k == 0 else 0, 12)]
    m = statistics.median(xs); tot += m
    print(f"{names[k]:16s} median {m:7.0f}  min {min(xs)} max {max(xs)}")
print("layer median sum", tot, " whole", sum(r))
