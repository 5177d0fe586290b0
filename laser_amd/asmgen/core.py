"""Instruction list builder: a tiny typed assembler front end shared by the text emitter and the interpreter."""


class Reg:
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        assert kind in ("v", "a", "s") and idx >= 0 and n >= 1
        assert not (kind in ("v", "a") and idx + n > 256), (kind, idx, n)
        assert not (kind == "s" and idx + n > 102), (kind, idx, n)
        self.kind, self.idx, self.n = kind, idx, n

    def __repr__(self):
        return f"{self.kind}{self.idx}" if self.n == 1 else f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def __getitem__(self, i):  # sub-register i of a tuple
        assert 0 <= i < self.n
        return Reg(self.kind, self.idx + i, 1)

    def sub(self, i, n):
        assert 0 <= i and i + n <= self.n
        return Reg(self.kind, self.idx + i, n)

    def regs(self):
        return [(self.kind, self.idx + i) for i in range(self.n)]


def v(i, n=1):
    return Reg("v", i, n)


def a(i, n=1):
    return Reg("a", i, n)


def s(i, n=1):
    return Reg("s", i, n)


class Sym:
    """vcc / exec / off / label operand (rendered verbatim)."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


VCC, EXEC, OFF, M0 = Sym("vcc"), Sym("exec"), Sym("off"), Sym("m0")


class Ins:
    __slots__ = ("op", "args", "mods", "comment")

    def __init__(self, op, args=(), mods=None, comment=None):
        self.op, self.args, self.mods, self.comment = op, list(args), dict(mods or {}), comment

    def text(self):
        if self.op == "label":
            return f"{self.args[0]}:"
        if self.op == "comment":
            return f"\t; {self.args[0]}"
        if self.op == "raw":
            return f"\t{self.args[0]}"
        if self.op == "s_waitcnt":
            parts = []
            if "vmcnt" in self.mods:
                parts.append(f"vmcnt({self.mods['vmcnt']})")
            if "lgkmcnt" in self.mods:
                parts.append(f"lgkmcnt({self.mods['lgkmcnt']})")
            t = "\ts_waitcnt " + " ".join(parts)
        else:
            def fmt(x):
                if isinstance(x, float):
                    assert x == 0.0 and not str(x).startswith("-"), "pass float literals as their bit pattern"
                    return "0"
                if isinstance(x, int) and not isinstance(x, bool) and (x > 64 or x < -16):
                    return hex(x & 0xffffffff)
                return x if isinstance(x, str) else repr(x)
            t = "\t" + self.op + (" " + ", ".join(fmt(x) for x in self.args) if self.args else "")
            for k, val in self.mods.items():
                if val is True:
                    t += f" {k}"
                elif val is not False and val is not None:
                    t += f" {k}:{val}"
        if self.comment:
            t += f"\t; {self.comment}"
        return t


class Prog:
    """An instruction list plus register allocators."""

    def __init__(self):
        self.ins = []
        self._v = 1  # v0 = workitem id
        self._s = 4  # s[0:1] kernarg pointer, s2 / s3 workgroup id x / y
        self._a = 0
        self._labels = 0
        self.names = {}

    # --- allocation ---
    def valloc(self, n=1, align=None, name=None):
        al = align or (1 if n == 1 else (2 if n == 2 else 4))
        self._v = (self._v + al - 1) // al * al
        r = Reg("v", self._v, n)
        self._v += n
        assert self._v <= 256, "out of arch VGPRs"
        if name:
            self.names[name] = r
        return r

    def salloc(self, n=1, align=None, name=None):
        al = align or (1 if n == 1 else (2 if n == 2 else 4))
        # the gaps that aligned tuples leave behind are handed out first (the large kernels use every SGPR there is)
        holes = self.__dict__.setdefault("_sholes", [])
        for h in sorted(holes):
            if h % al == 0 and all(h + k in holes for k in range(n)):
                for k in range(n):
                    holes.remove(h + k)
                r = Reg("s", h, n)
                if name:
                    self.names[name] = r
                return r
        start = (self._s + al - 1) // al * al
        holes.extend(range(self._s, start))
        self._s = start
        r = Reg("s", self._s, n)
        self._s += n
        assert self._s <= 100, "out of SGPRs"
        if name:
            self.names[name] = r
        return r

    def aalloc(self, n=1, name=None):
        al = 1 if n == 1 else (2 if n == 2 else 4)
        self._a = (self._a + al - 1) // al * al
        r = Reg("a", self._a, n)
        self._a += n
        assert self._a <= 256, "out of AGPRs"
        if name:
            self.names[name] = r
        return r

    def label(self, hint="L"):
        self._labels += 1
        return f".L_{hint}_{self._labels}"

    # --- emission ---
    def emit(self, op, *args, comment=None, **mods):
        i = Ins(op, args, mods, comment)
        self.ins.append(i)
        return i

    def place(self, name):
        self.ins.append(Ins("label", [name]))

    def note(self, text):
        self.ins.append(Ins("comment", [text]))

    def text(self):
        return "\n".join(i.text() for i in self.ins) + "\n"
