"""VALU instructions of one kernel attributed to source lines (development aid).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -S --cuda-device-only -Iinclude \
        -o /tmp/kg.s stm32_speech_recognition_amd/csrc/k_mfcc.hip
  python profiles/isa_by_line.py /tmp/kg.s k_mfcc [.LBB0_44 ...]     (restrict to basic blocks)
"""
import collections
import re
import sys


def main():
    path, kern, only = sys.argv[1], sys.argv[2], set(sys.argv[3:])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN2sr\d+" + kern + r"[EI]", l))
    cur, blk = None, "entry"
    cnt = collections.Counter()
    ops = collections.defaultdict(collections.Counter)
    for l in lines[start + 1:]:
        if l.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blk = m.group(1)
            continue
        m = re.match(r"\s+\.loc\s+\d+\s+(\d+)\s+(\d+)", l)
        if m:
            cur = int(m.group(1))
            continue
        m = re.match(r"^\s+(v_[a-z_0-9]+)", l)
        if m and (not only or blk in only):
            cnt[cur] += 1
            ops[cur][m.group(1)] += 1
    src = open("stm32_speech_recognition_amd/csrc/k_mfcc.hip").read().split("\n")
    for ln, c in sorted(cnt.items()):
        top = " ".join(f"{k}:{v}" for k, v in ops[ln].most_common(4))
        print(f"{ln:5d} {c:4d}  {src[ln - 1].strip()[:70]:70s} | {top}")
    print("total", sum(cnt.values()))


if __name__ == "__main__":
    main()
