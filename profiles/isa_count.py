"""Static per-basic-block instruction counts of one kernel in hipcc -S output (development aid).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -Iinclude \
        -o /tmp/k.s stm32_speech_recognition_amd/csrc/k_mfcc.hip
  python profiles/isa_count.py /tmp/k.s k_mfcc
"""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2sr\d+" + kern + r"[EI]", l))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = collections.Counter()
    for l in lines[start + 1:]:
        if l.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = collections.Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        blocks[cur][cls] += 1
        blocks[cur]["op:" + op] += 1
    tot = collections.Counter()
    for b, c in blocks.items():
        n = sum(v for k, v in c.items() if not k.startswith("op:"))
        if n >= 20:
            print(f"{b:12s} valu {c['valu']:5d} salu {c['salu']:4d} lds {c['lds']:4d} vmem {c['vmem']:4d}")
        tot.update(c)
    print("total        valu %d salu %d lds %d vmem %d" % (tot["valu"], tot["salu"], tot["lds"], tot["vmem"]))
    if len(sys.argv) > 3:
        b = blocks[sys.argv[3]]
        for k, v in sorted(((k, v) for k, v in b.items() if k.startswith("op:")), key=lambda kv: -kv[1])[:40]:
            print(f"   {k[3:]:28s} {v}")


if __name__ == "__main__":
    main()
