"""TEST INFRASTRUCTURE: runs the reference's ARM assembly FFT *from its own source text*.

No ARM toolchain or emulator exists in the build image, so `cr4_fft_1024_stm32.s` (armasm syntax, Thumb-2, Cortex-M3)
cannot be assembled or executed natively.  This module is a small assembler + interpreter for exactly the subset of
armasm / Thumb-2 that file uses; it reads the file where it lies under /root/reference (nothing is copied), expands its
macros, lays out its DCW coefficient table and executes `cr4_fft_1024_stm32` on a simulated memory.  Its only job is to
pin `oracle/q15_fft.c` (the C restatement every other test relies on) against the reference's own instructions:
`tests/test_oracle.py::test_asm_fft_interpreted_equals_restatement` and `tests/golden/make_golden.py` use it.

Supported, with the semantics of the ARMv7-M Architecture Reference Manual (all arithmetic modulo 2^32):
  directives   THUMB REQUIRE8 PRESERVE8 AREA EXPORT EXTERN END, `name RN Rk`, `name EQU expr`, MACRO/MEND with
               $parameters (nested invocations), DCW, labels in column 0, `;` comments
  data moves   MOV{S} Rd, #imm | Rm{, shift}          shift = LSL|LSR|ASR #expr
  arithmetic   ADD|SUB{S}{cond} Rd, Rn, #imm | Rm{, shift}   (two-operand form ADD Rd, #imm), MUL, MLA, RBIT, CMP
  memory       LDRSH Rt, [Rn{, #imm}]   STRH Rt, [Rn{, #imm}]   STRH Rt, [Rn], #imm   STMFD SP!, {..}   LDMFD SP!, {..}
  control      B{NE,GE} label, IT NE (the following instruction carries its own condition suffix), ADRL Rd, label
Flags: N, Z from the result; C = shifter carry-out for MOVS with a shift, NOT(borrow) for SUBS/CMP; V = signed
overflow for SUBS/CMP.  A pop into PC ends the run.
"""
import re

M32 = 0xFFFFFFFF
DEFAULT_PATH = "/root/reference/Src/BSP/cr4_fft_1024_stm32.s"
_IGNORED = {"THUMB", "REQUIRE8", "PRESERVE8", "AREA", "EXPORT", "EXTERN", "END"}
_CONDS = ("NE", "GE", "EQ", "LT")


def _split_operands(text):
    """split on commas that are not inside [] or {}"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "[{":
            depth += 1
        elif ch in "]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class AsmFft:
    def __init__(self, path=DEFAULT_PATH):
        self.regs_alias = {f"R{i}": i for i in range(16)}
        self.regs_alias.update(SP=13, LR=14, PC=15)
        self.equ = {}
        self.macros = {}
        self.prog = []          # (mnemonic, [operands], source line number)
        self.labels = {}        # label -> program index
        self.data = bytearray() # DCW words, little endian
        self.data_labels = {}   # label -> offset into data
        self._assemble(open(path, encoding="latin-1").read().splitlines())

    # ---------------------------------------------------------------- assembler
    def _assemble(self, lines):
        cleaned = []
        for no, raw in enumerate(lines, 1):
            line = raw.split(";")[0].rstrip()
            if line.strip():
                cleaned.append((no, line))
        i = 0
        pending_label = None
        while i < len(cleaned):
            no, line = cleaned[i]
            toks = line.split()
            if toks[0] == "MACRO":                      # definition: header on the next line, body until MEND
                hdr = cleaned[i + 1][1].split(None, 1)
                name, params = hdr[0], [p.strip() for p in (hdr[1].split(",") if len(hdr) > 1 else [])]
                body = []
                i += 2
                while cleaned[i][1].split()[0] != "MEND":
                    body.append(cleaned[i])
                    i += 1
                self.macros[name] = (params, body)
                i += 1
                continue
            if len(toks) >= 3 and toks[1] == "RN":
                self.regs_alias[toks[0]] = int(toks[2][1:])
            elif len(toks) >= 3 and toks[1] == "EQU":
                self.equ[toks[0]] = self._expr(" ".join(toks[2:]))
            elif toks[0] in _IGNORED:
                pass
            elif not line[0].isspace():                 # label in column 0 (possibly followed by nothing)
                pending_label = toks[0]
                if len(toks) > 1:
                    self._emit(no, " ".join(toks[1:]), pending_label)
                    pending_label = None
            else:
                self._emit(no, line.strip(), pending_label)
                pending_label = None
            i += 1

    def _emit(self, no, text, label, subst=None):
        if subst:
            for k, v in subst.items():                  # longest names first so $pssDin / $pssDout / $pssK do not clash
                text = text.replace(k, v)
        parts = text.split(None, 1)
        mnem = parts[0]
        ops = _split_operands(parts[1]) if len(parts) > 1 else []
        if mnem == "DCW":
            if label:
                self.data_labels[label] = len(self.data)
            for o in ops:
                v = self._expr(o) & 0xFFFF
                self.data += bytes((v & 0xFF, v >> 8))
            return
        if label:
            self.labels[label] = len(self.prog)
        if mnem in self.macros:
            params, body = self.macros[mnem]
            assert len(ops) == len(params), (no, text)
            sub = dict(sorted(zip(params, ops), key=lambda kv: -len(kv[0])))
            for bno, bline in body:
                self._emit(bno, bline.strip(), None, sub)
            return
        self.prog.append((mnem.upper(), ops, no))

    def _expr(self, text):
        t = text.strip().lstrip("#").strip()
        t = re.sub(r"0x[0-9a-fA-F]+", lambda m: str(int(m.group(0), 16)), t)
        t = re.sub(r"[A-Za-z_]\w*", lambda m: str(self.equ[m.group(0)]), t)
        assert re.fullmatch(r"[0-9+\-*<>() ]+", t), text
        return int(eval(t, {"__builtins__": {}}))

    # ---------------------------------------------------------------- machine
    def _reg(self, name):
        return self.regs_alias[name.strip()] if name.strip() in self.regs_alias else self.regs_alias[name.strip().upper()]

    def _is_reg(self, name):
        n = name.strip()
        return n in self.regs_alias or n.upper() in self.regs_alias

    @staticmethod
    def _shift(val, kind, amt):
        """returns (result, carry_out or None)"""
        kind = kind.upper()
        if amt == 0:
            return val & M32, None
        if kind == "LSL":
            return (val << amt) & M32, (val >> (32 - amt)) & 1 if amt <= 32 else 0
        if kind == "LSR":
            return (val >> amt) & M32 if amt < 32 else 0, (val >> (amt - 1)) & 1 if amt <= 32 else 0
        if kind == "ASR":
            s = val - (1 << 32) if val & 0x80000000 else val
            amt = min(amt, 32)
            return (s >> amt) & M32, (s >> (amt - 1)) & 1
        raise ValueError(kind)

    def _operand2(self, r, ops):
        """flexible second operand: '#imm' | 'Rm' | 'Rm', 'LSL#k' (k may be an expression)"""
        if ops[0].startswith("#"):
            return self._expr(ops[0]) & M32, None
        val = r[self._reg(ops[0])]
        if len(ops) > 1:
            m = re.match(r"(LSL|LSR|ASR)\s*#(.+)", ops[1].strip(), re.I)
            return self._shift(val, m.group(1), self._expr(m.group(2)))
        return val, None

    def run(self, words):
        """words: 1024 packed complex samples (re = low s16, im = high s16) -> 1024 output words"""
        IN, OUT, STACK, TABLE, RET = 0x20000000, 0x20002000, 0x20008000, 0x08001000, 0xFFFFFFF0
        mem = {}

        def rd16(a):
            return mem.get(a, 0) | (mem.get(a + 1, 0) << 8)

        def wr16(a, v):
            mem[a] = v & 0xFF
            mem[a + 1] = (v >> 8) & 0xFF

        def rd32(a):
            return rd16(a) | (rd16(a + 2) << 16)

        def wr32(a, v):
            wr16(a, v & 0xFFFF)
            wr16(a + 2, (v >> 16) & 0xFFFF)

        for i, w in enumerate(words):
            wr32(IN + 4 * i, int(w) & M32)
        for i, b in enumerate(self.data):
            mem[TABLE + i] = b
        r = [0] * 16
        r[0], r[1], r[2], r[13], r[14] = OUT, IN, 1024, STACK, RET
        N = Z = Cf = V = 0
        pc = self.labels["cr4_fft_1024_stm32"]
        steps = 0
        while True:
            mnem, ops, no = self.prog[pc]
            pc += 1
            steps += 1
            assert steps < 5_000_000, "runaway"
            cond = None
            base = mnem
            for c in _CONDS:
                if mnem.endswith(c) and mnem[:-2] in ("B", "SUB", "ADD", "MOV"):
                    cond, base = c, mnem[:-2]
            if cond is not None:
                ok = {"NE": Z == 0, "EQ": Z == 1, "GE": N == V, "LT": N != V}[cond]
                if not ok:
                    continue
            setf = base.endswith("S") and base[:-1] in ("MOV", "SUB", "ADD")
            if setf:
                base = base[:-1]
            if base == "IT":
                continue
            if base == "B":
                pc = self.labels[ops[0]]
            elif base == "MOV":
                val, c = self._operand2(r, ops[1:])
                r[self._reg(ops[0])] = val
                if setf:
                    N, Z = val >> 31, int(val == 0)
                    if c is not None:
                        Cf = c
            elif base in ("ADD", "SUB"):
                if len(ops) == 2:                       # ADD Rd, #imm
                    ops = [ops[0], ops[0], ops[1]]
                a = r[self._reg(ops[1])]
                b, _ = self._operand2(r, ops[2:])
                if base == "ADD":
                    res = (a + b) & M32
                    if setf:
                        Cf = int(a + b > M32)
                        V = int(((a ^ res) & (b ^ res)) >> 31)
                else:
                    res = (a - b) & M32
                    if setf:
                        Cf = int(a >= b)
                        V = int(((a ^ b) & (a ^ res)) >> 31)
                if setf:
                    N, Z = res >> 31, int(res == 0)
                r[self._reg(ops[0])] = res
            elif base == "CMP":
                a = r[self._reg(ops[0])]
                b, _ = self._operand2(r, ops[1:])
                res = (a - b) & M32
                N, Z, Cf, V = res >> 31, int(res == 0), int(a >= b), int(((a ^ b) & (a ^ res)) >> 31)
            elif base == "MUL":
                r[self._reg(ops[0])] = (r[self._reg(ops[1])] * r[self._reg(ops[2])]) & M32
            elif base == "MLA":
                r[self._reg(ops[0])] = (r[self._reg(ops[1])] * r[self._reg(ops[2])] + r[self._reg(ops[3])]) & M32
            elif base == "RBIT":
                r[self._reg(ops[0])] = int(f"{r[self._reg(ops[1])]:032b}"[::-1], 2)
            elif base == "ADRL":
                r[self._reg(ops[0])] = TABLE + self.data_labels[ops[1]]
            elif base in ("LDRSH", "STRH"):
                m = re.fullmatch(r"\[\s*([^,\]]+)\s*(?:,\s*(#[^\]]+))?\]", ops[1].strip())
                rn = self._reg(m.group(1))
                addr = (r[rn] + (self._expr(m.group(2)) if m.group(2) else 0)) & M32
                if base == "LDRSH":
                    v = rd16(addr)
                    r[self._reg(ops[0])] = (v - 0x10000) & M32 if v & 0x8000 else v
                else:
                    wr16(addr, r[self._reg(ops[0])] & 0xFFFF)
                if len(ops) == 3:                       # post-index
                    r[rn] = (r[rn] + self._expr(ops[2])) & M32
            elif base in ("STMFD", "LDMFD"):
                assert ops[0].replace(" ", "").upper() == "SP!"
                names = []
                for part in ops[1].strip("{} ").split(","):
                    part = part.strip()
                    if "-" in part and not self._is_reg(part):
                        lo, hi = part.split("-")
                        names += list(range(self._reg(lo), self._reg(hi) + 1))
                    else:
                        names.append(self._reg(part))
                names = sorted(set(names))
                if base == "STMFD":
                    r[13] = (r[13] - 4 * len(names)) & M32
                    for k, reg in enumerate(names):
                        wr32(r[13] + 4 * k, r[reg])
                else:
                    done = False
                    for k, reg in enumerate(names):
                        v = rd32(r[13] + 4 * k)
                        if reg == 15:
                            assert v == RET, hex(v)
                            done = True
                        else:
                            r[reg] = v
                    r[13] = (r[13] + 4 * len(names)) & M32
                    if done:
                        break
            else:
                raise NotImplementedError(f"line {no}: {mnem} {ops}")
        return [rd32(OUT + 4 * i) for i in range(1024)], steps
