#!/usr/bin/env python3
"""A dataflow interpreter for the structural VHDL of the reference's arithmetic entities -- TEST INFRASTRUCTURE, this container only.

    python tools/rtl_interp.py [--per-case N] [--seed S]          (needs /root/reference; reads it, copies nothing)

What it is for.  oracle/dsp48_twin.py wires every multiplier / complex multiplier / adder entity "as its port map reads" -- by hand.
This tool removes the hand from that step: it PARSES the reference's own files (src/vhdl/math/mults/*.vhd, src/vhdl/math/cmult/*.vhd,
src/vhdl/math/int_addsub_dsp48.vhd, src/vhdl/fft/int_dif2_fly.vhd, int_dit2_fly.vhd, src/vhdl/twiddle/row_twiddle_tay.vhd, rom_twiddle_int.vhd; the generate loops of src/vhdl/fft/int_fftNk.vhd / int_ifftNk.vhd for the stage
schedule), elaborates an entity for given generics (its functions -- constants by XSER, MATH_PI arithmetic, the ROM a loop fills --, if / for generate,
local signals, entity instantiations with generic / port maps) and evaluates it as a dataflow network: every concurrent signal
assignment (slices, SXT, (others => x), single bits, literals; `when rising_edge(clk)` and `after ...` are delays and are ignored), the
clocked processes of the butterflies (if / else on a bit, `+ '1'`, `not`) as the combinational functions they register, every
DSP48E1 / DSP48E2 instance through the ONE thing that is not in the reference -- the slice model of oracle/dsp48_twin.py (`dsp48`,
from UG479 / UG579) -- and every `entity work.x` instance recursively.  The results are compared with the hand-wired twin and with
oracle_py on random and corner operands: a port map the twin (or the oracles) read wrongly shows as a mismatch against the text itself.

The delay lines (int_delay_line.vhd, int_delay_wrap.vhd: counters, two memories, a crossbar) are the one place where time IS the function:
those two entities are CLOCKED (CycleSim: every register, counter, memory and clocked process of the text, cycle by cycle) on whole frames.

It is NOT a VHDL simulator (no delta cycles, no delays, no resolution, std_logic as bits; everything else is evaluated as a dataflow network) and it does not make the reference "buildable":
parity stays unpinned.  It runs where /root/reference exists; tests/test_rtl_interp.py skips elsewhere.  Nothing it reads is stored.
"""
from __future__ import annotations

import os
import random
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import dsp48_twin as tw  # noqa: E402

REF = os.environ.get("INTFFT_REFERENCE", "/root/reference")
DIRS = ["src/vhdl/math/mults", "src/vhdl/math/cmult", "src/vhdl/math", "src/vhdl/fft", "src/vhdl/twiddle", "src/vhdl/delay", "src/vhdl/buffers",
        "src/vhdl/main"]


def available() -> bool:
    return os.path.isdir(os.path.join(REF, DIRS[0]))


# ------------------------------------------------------------------------------------------------ parsing

def _load(entity: str) -> str:
    for d in DIRS:
        p = os.path.join(REF, d, entity + ".vhd")
        if not os.path.exists(p):  # file names keep the entity's capitals (int_fftNk.vhd); entity names are lower case in here
            hit = [f for f in os.listdir(os.path.join(REF, d)) if f.lower() == entity + ".vhd"]
            p = os.path.join(REF, d, hit[0]) if hit else p
        if os.path.exists(p):
            txt = open(p, encoding="latin-1").read()
            txt = re.sub(r"--[^\n]*", "", txt)          # comments
            return re.sub(r"\s+", " ", txt).lower()     # VHDL is case-insensitive: everything lower case, one line
    raise FileNotFoundError(entity)


def _split_top(s: str, sep: str):
    """split on sep outside parentheses"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


FUNC = re.compile(r"function (\w+) ?(?:\((.*?)\))? ?return \w+ is(.*?)\bbegin (.*?)end (?:function(?: \1)?|\1) ?;")


def _take_functions(text: str, table: dict) -> str:
    """cut `function f(params) return t is <variables> begin <body> end [function] f;` out of a declarative region into `table`"""
    for m in FUNC.finditer(text):
        params = [nm.strip() for x in m.group(2).split(";") for nm in x.split(":")[0].split(",")] if m.group(2) else []
        table[m.group(1)] = (params, _parse_fn(m.group(4)))
    return FUNC.sub(" ", text)


def _parse_fn(text: str):
    """function body: ("set", target, expr) | ("if", [(cond, seq)], else) | ("for", var, lo, hi, seq) | ("return", expr)"""
    toks = [t.strip() for t in re.split(r"(\bend if ?;|\bend loop ?;|\belsif\b|\belse\b|\bif\b|\bthen\b|\bloop\b|;)", text) if t.strip()]
    pos = 0

    def seq(stop):
        nonlocal pos
        out = []
        while pos < len(toks) and not any(toks[pos].replace(" ", "") == x.replace(" ", "") for x in stop):
            t = toks[pos]
            if t == "if":
                branches, els = [], []
                cond = toks[pos + 1]
                pos += 3
                branches.append((cond, seq(("elsif", "else", "end if;"))))
                while toks[pos] == "elsif":
                    cond = toks[pos + 1]
                    pos += 3
                    branches.append((cond, seq(("elsif", "else", "end if;"))))
                if toks[pos] == "else":
                    pos += 1
                    els = seq(("end if;",))
                pos += 1
                out.append(("if", branches, els))
            elif re.match(r"(\w+ ?: ?)?for ", t):   # `xl: for ii in ...`: a labelled loop (iobuf_wrap_int2)
                m = re.match(r"(?:\w+ ?: ?)?for (\w+) in (.*) to (.*)$", t)
                pos += 2  # the header, `loop`
                out.append(("for", m.group(1), m.group(2), m.group(3), seq(("end loop;",))))
                pos += 1
            elif t == ";":
                pos += 1
            elif t.startswith("return "):
                out.append(("return", t[7:]))
                pos += 1
            else:
                m = re.match(r"(\w+(?: ?\(.*?\))?) ?:= ?(.*)$", t)
                assert m, "unparsed function statement: %r" % t[:100]
                out.append(("set", m.group(1).strip(), m.group(2).strip()))
                pos += 1
        return out

    return seq(())


def _vint(x):
    """VHDL INTEGER(real): round to nearest"""
    import math
    return int(math.floor(x + 0.5))


def _fn_eval(expr: str, env: dict):
    import math
    e = re.sub(r"\b0+(\d)", r"\1", expr)
    scope = dict(env, math_pi=math.pi, integer=_vint, conv_std_logic_vector=lambda v, n: int(v) & ((1 << n) - 1), true=True, false=False,
                 real=float, cos=math.cos, sin=math.sin, conv_signed=lambda v, n: int(v) & ((1 << n) - 1), std_logic_vector=lambda v: v)
    return eval(e, {"__builtins__": {}}, scope)  # noqa: S307


def call_function(table: dict, name: str, args: list, env: dict):
    params, ast = table[name]
    scope = dict(env)
    scope.update(zip(params, args))

    def run(seq):
        for node in seq:
            if node[0] == "set":
                m = re.match(r"(\w+) ?\((.*?)\) ?\((.*) downto (.*)\)$", node[1])
                if m:  # arr(i)(hi downto lo) := vector
                    arr, idx = scope.setdefault(m.group(1), {}), int(_fn_eval(m.group(2), scope))
                    hi, lo = int(_fn_eval(m.group(3), scope)), int(_fn_eval(m.group(4), scope))
                    arr[idx] = arr.get(idx, 0) | ((int(_fn_eval(node[2], scope)) & ((1 << (hi - lo + 1)) - 1)) << lo)
                    continue
                m = re.match(r"(\w+) ?\((.*)\)$", node[1])
                if m:
                    scope.setdefault(m.group(1), {})[int(_fn_eval(m.group(2), scope))] = _fn_eval(node[2], scope)
                else:
                    scope[node[1]] = _fn_eval(node[2], scope)
            elif node[0] == "if":
                for cond, body in node[1]:
                    if _cond(cond, scope):
                        r = run(body)
                        if r is not None:
                            return r
                        break
                else:
                    r = run(node[2])
                    if r is not None:
                        return r
            elif node[0] == "for":
                for v in range(int(_fn_eval(node[2], scope)), int(_fn_eval(node[3], scope)) + 1):
                    scope[node[1]] = v
                    r = run(node[4])
                    if r is not None:
                        return r
            else:
                return ("ret", _fn_eval(node[1], scope))
        return None

    r = run(ast)
    return r[1]


GENERATE = re.compile(r"generate\b")
PROCESS = re.compile(r"process\b")


def _balanced(s: str) -> bool:
    d = 0
    for ch in s:
        d += ch == "("
        d -= ch == ")"
        if d < 0:
            return False
    return d == 0


def _split_kw(s: str, kw: str):
    """split on a keyword (with its spaces) outside parentheses"""
    out, depth, cur, i = [], 0, "", 0
    while i < len(s):
        if depth == 0 and s.startswith(kw, i):
            out.append(cur.strip())
            cur = ""
            i += len(kw)
            continue
        depth += s[i] == "("
        depth -= s[i] == ")"
        cur += s[i]
        i += 1
    out.append(cur.strip())
    return out


class Entity:
    def __init__(self, name: str):
        self.name = name
        t = _load(name)
        m = re.search(r"entity %s is(.*?)end %s ?;" % (name, name), t)
        head = m.group(1)
        self.generics = []  # (name, default text)
        g = re.search(r"generic ?\((.*?)\) ?; ?port", head)
        if g:
            for item in _split_top(g.group(1), ";"):
                mm = re.match(r"(\w+) ?: ?\w+ ?(?::= ?(.*))?$", item)
                self.generics.append((mm.group(1), (mm.group(2) or "").strip()))
        p = re.search(r"port ?\((.*)\) ?;", head)
        self.ports = {}  # name -> (dir, range text or None)
        for item in _split_top(p.group(1), ";"):
            mm = re.match(r"(\w+) ?: ?(in|out) +std_logic(?:_vector ?\((.*)\))?$", item)
            self.ports[mm.group(1)] = (mm.group(2), mm.group(3))
        a = re.search(r"architecture (\w+) of %s is(.*)end (?:%s|\1) ?;" % (name, name), t)  # `end <architecture name>;`: ramb_tdp_rw
        body = a.group(2)
        self.functions = {}
        body = _take_functions(body, self.functions)
        i = body.index(" begin ")
        self.decls, self.body = body[:i], body[i + 7:]


def _statements(s: str):
    """top-level statements of a concurrent region: yields the text of each `...;` with generate blocks kept whole"""
    out, depth, cur, i, n = [], 0, "", 0, len(s)
    gen = 0
    while i < n:
        m = GENERATE.match(s, i) if (i == 0 or not (s[i - 1].isalnum() or s[i - 1] == "_")) else None
        pm = PROCESS.match(s, i) if (m is None and (i == 0 or not (s[i - 1].isalnum() or s[i - 1] == "_"))) else None
        if pm:
            gen += -1 if cur.rstrip().endswith("end") else 1
            cur += "process"
            i = pm.end()
            continue
        if m:
            # `end generate` closes, a bare `generate` opens
            if cur.rstrip().endswith("end"):
                gen -= 1
            else:
                gen += 1
            cur += "generate"
            i = m.end()
            continue
        ch = s[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == ";" and depth == 0 and gen == 0:
            if cur.strip():
                out.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def _parse_seq(text: str):
    """sequential statements of a clocked process: [("assign", lhs, rhs) | ("if", [(cond, seq), ...], else_seq)]"""
    toks = re.split(r"(\bend if ?;|\belsif\b|\belse\b|\bif\b|\bthen\b|;)", text)
    toks = [t.strip() for t in toks if t.strip()]
    pos = 0

    def seq(stop):
        nonlocal pos
        out = []
        while pos < len(toks) and toks[pos] not in stop:
            t = toks[pos]
            if t == "if":
                branches, els = [], []
                pos += 1
                cond = toks[pos]
                pos += 2  # cond, then
                branches.append((cond, seq(("elsif", "else", "end if;", "end if ;"))))
                while toks[pos] == "elsif":
                    cond = toks[pos + 1]
                    pos += 3
                    branches.append((cond, seq(("elsif", "else", "end if;", "end if ;"))))
                if toks[pos] == "else":
                    pos += 1
                    els = seq(("end if;", "end if ;"))
                pos += 1  # end if;
                out.append(("if", branches, els))
            elif t == ";":
                pos += 1
            else:
                m = re.match(r"([\w]+(?: ?\(.*?\))?) ?(?:<|:)= ?(.*)$", t)  # `:=`: a shared variable used as a memory (buffers/)
                assert m, "unparsed sequential statement: %r" % t[:100]
                out.append(("assign", m.group(1).strip(), re.sub(r"\s*\bafter [\w.]+( ns\b)?", "", m.group(2)).strip()))
                pos += 1
        return out

    return seq(())


# ------------------------------------------------------------------------------------------------ evaluation

class NotReady(Exception):
    pass


class Sig:
    __slots__ = ("hi", "lo", "val", "known")

    def __init__(self, hi, lo):
        self.hi, self.lo, self.val, self.known = hi, lo, 0, 0

    @property
    def width(self):
        return self.hi - self.lo + 1

    def full(self):
        return self.width <= 0 or self.known == (1 << self.width) - 1

    def put(self, hi, lo, v):
        if hi < lo:
            return  # a null range (N_INV = 0 in int_delay_wrap): no bits
        assert self.lo <= lo <= hi <= self.hi, "slice (%d downto %d) outside (%d downto %d): does not elaborate" % (hi, lo, self.hi, self.lo)
        w = hi - lo + 1
        m = ((1 << w) - 1) << (lo - self.lo)
        assert not (self.known & m), "double driver"
        self.val |= (v & ((1 << w) - 1)) << (lo - self.lo)
        self.known |= m

    def get_known(self, hi, lo):
        if hi < lo:
            return True
        w = hi - lo + 1
        m = ((1 << w) - 1) << (lo - self.lo)
        return (self.known & m) == m

    def get(self, hi, lo):
        if hi < lo:
            return 0  # a null range reads as the empty vector (conv_integer of it is 0)
        assert self.lo <= lo <= hi <= self.hi, "slice (%d downto %d) outside (%d downto %d): does not elaborate" % (hi, lo, self.hi, self.lo)
        w = hi - lo + 1
        m = ((1 << w) - 1) << (lo - self.lo)
        if (self.known & m) != m:
            raise NotReady()
        return (self.val >> (lo - self.lo)) & ((1 << w) - 1)


def _int(expr: str, env: dict) -> int:
    e = re.sub(r"\b0+(\d)", r"\1", expr)  # "00" -> "0"
    return int(eval(e, {"__builtins__": {}}, env))  # noqa: S307 -- integer index arithmetic of the parsed generics


def _cond(expr: str, env: dict) -> bool:
    e = re.sub(r"(?<![<>/=])=(?!=)", "==", expr).replace("/==", "!=")
    return bool(eval(e, {"__builtins__": {}}, dict(env, true=True, false=False)))  # noqa: S307


class Inst:
    """One elaborated entity: its signals and the flat list of pending statements."""

    def __init__(self, ent: Entity, generics: dict):
        self.ent = ent
        self.env = {}
        for g, dflt in ent.generics:
            v = generics.get(g, dflt.strip('"') if dflt.startswith('"') else (int(dflt) if dflt.lstrip("-").isdigit() else
                                {"true": True, "false": False}.get(dflt)))
            self.env[g] = v
        self.sigs = {}
        self.alias = {}    # delay lines: name -> the signal every tap carries
        self.array_types = {}  # array type -> word width
        self.arrays = {}   # memories: name -> (word width, {index: value})
        self.consts = set()  # vector constants: signals that keep their value from one evaluation to the next
        self.funcs = dict(ent.functions)
        self.subs = {}     # elaborated sub-entities, by position in `pending` (an elaboration is reused from one evaluation to the next)
        self.pending = []  # ("assign", lhs, rhs, env) | ("inst", label, unit, gmap, pmap, env)
        self._decls(ent.decls, self.env)
        for p, (_, rng) in ent.ports.items():
            if rng is None:
                self.sigs[p] = Sig(0, 0)
            else:
                hi, lo = rng.split(" downto ")
                self.sigs[p] = Sig(_int(hi, self.env), _int(lo, self.env))
        self._region(ent.body, dict(self.env))

    def _decls(self, text, env):
        for st in _split_top(text, ";"):
            m = re.match(r"constant (\w+) ?: ?(\w+)(?: ?\((.*?) downto (.*?)\))? ?:= ?(.*)$", st)
            if m:
                name, init = m.group(1), m.group(5).strip()
                k = re.match(r"(\w+)(?: ?\((.*)\))?$", init)
                try:
                    if k and k.group(1) in self.funcs:  # constant awd : natural := find_widtha(xser) | := find_widtha | := read_rom(ii)
                        args = [env[a.strip()] if a.strip() in env else _int(a, env) for a in k.group(2).split(",")] if k.group(2) else []
                        env[name] = call_function(self.funcs, k.group(1), args, env)
                    elif m.group(3):  # a vector constant: std_logic_vector(conv_unsigned(x, n)) and the like
                        kk = re.match(r"std_logic_vector ?\( ?conv_unsigned ?\((.*), ?(\d+) ?\) ?\)$", init)
                        self.sigs[name] = Sig(_int(m.group(3), env), _int(m.group(4), env))
                        self.sigs[name].put(self.sigs[name].hi, self.sigs[name].lo, _int(kk.group(1), env))
                        self.consts.add(name)
                        continue
                    else:
                        env[name] = _int(init, env)
                    self.env.setdefault(name, env[name])
                except Exception:  # a delay constant built from functions this tool has no use for: timing, not arithmetic
                    pass
                continue
            if st.startswith("signal "):
                st = re.sub(r" ?:= ?.*$", "", st)  # initial values: everything starts at 0 / unknown here
            m = re.match(r"signal ([\w, ]+) ?: ?std_logic(?:_vector ?\((.*) downto (.*)\))?$", st)
            if m:
                try:
                    rng = (_int(m.group(2), env), _int(m.group(3), env)) if m.group(2) else (0, 0)
                except Exception:  # a range in terms of a delay constant this tool does not evaluate: a control signal, never read here
                    continue
                for nm in m.group(1).split(","):
                    self.sigs[nm.strip()] = Sig(*rng)
                continue
            m = re.match(r"type (\w+) is array ?\(.*\) of std_logic_vector ?\((.*) downto (.*)\)$", st)
            if m:
                try:
                    self.array_types[m.group(1)] = _int(m.group(2), env) - _int(m.group(3), env) + 1
                except Exception:
                    pass
                continue
            m = re.match(r"signal ([\w, ]+) ?: ?(\w+)$", st)
            if m and m.group(2) in self.array_types:  # a memory: word index -> value (the cycle simulator reads and writes it)
                for nm in m.group(1).split(","):
                    self.arrays[nm.strip()] = (self.array_types[m.group(2)], {})
            # anything else (delay constants, other types) is timing, not arithmetic: ignored

    def _gen_body(self, inner, env):
        """a generate body: [declarations] [begin] statements"""
        if re.match(r" ?(signal|constant|type|function) ", inner):
            inner = _take_functions(inner, self.funcs)
            k = re.match(r"(.*?)\bbegin (.*)$", inner)
            self._decls(k.group(1), env)
            return k.group(2)
        return re.sub(r"^ ?begin ", "", inner)

    def _region(self, text, env):
        for st in _statements(text):
            m = re.match(r"(\w+) ?: ?if (.*?) generate (.*) end generate(?: \w+)?$", st)
            if m:
                if _cond(m.group(2), env):
                    self._region(self._gen_body(m.group(3), env), env)
                continue
            m = re.match(r"(\w+) ?: ?for (\w+) in (.*?) to (.*?) generate (.*) end generate(?: \w+)?$", st)
            if m:
                for v in range(_int(m.group(3), env), _int(m.group(4), env) + 1):
                    e2 = dict(env, **{m.group(2): v})
                    self._region(self._gen_body(m.group(5), e2), e2)
                continue
            m = re.match(r"(\w+) ?: ?(entity work\.\w+|dsp48e1|dsp48e2) ?(?:generic map ?\((.*?)\) ?)?port map ?\((.*)\)$", st)
            if m:
                gmap = dict(x.split("=>", 1) for x in _split_top(m.group(3), ",")) if m.group(3) else {}
                pmap = dict(x.split("=>", 1) for x in _split_top(m.group(4), ","))
                self.pending.append(("inst", m.group(1), m.group(2).replace("entity work.", ""), {k.strip(): v.strip() for k, v in gmap.items()},
                                     {k.strip(): v.strip() for k, v in pmap.items()}, env))
                continue
            m = re.match(r"(\w+) ?: ?process ?\(.*?\) ?is begin (.*) end process(?: \w+)?$", st)
            if m:
                body = m.group(2).strip()
                k = re.match(r"if (?:rising_edge ?\( ?clk ?\)|\( ?clk'event and clk ?= ?'1' ?\)) then (.*) end if ?;?$", body)
                self.pending.append(("proc", _parse_seq(k.group(1) if k else body), env))
                continue
            m = re.match(r"(\w+) ?<= ?\1 ?\(.*? downto 0 ?\) ?& ?(\w+)(?: when rising_edge ?\( ?clk ?\))?$", st)
            if m:  # a delay line x <= x(n-2 downto 0) & y: every tap of x is y, some clocks later
                self.alias[m.group(1)] = m.group(2)
                continue
            m = re.match(r"([\w]+(?: ?\(.*?\))?) ?<= ?(.*)$", st)
            assert m, "unparsed statement: %r" % st[:120]
            rhs = re.sub(r"\s*\bafter [\w.]+( ns\b)?", "", m.group(2))
            reg = bool(re.search(r"\bwhen rising_edge ?\( ?clk ?\)", rhs))
            en = re.search(r"\bwhen rising_edge ?\( ?clk ?\) and (.*)$", rhs)  # a register with a clock enable
            rhs = re.sub(r"\s*\bwhen rising_edge ?\( ?clk ?\)( and .*)?$", "", rhs).strip()
            self.pending.append(("assign", m.group(1).strip(), rhs, env, reg, en.group(1).strip() if en else None))

    # ---- values -------------------------------------------------------------------------------------------------------------------
    def _ref(self, text, env):
        """name | name(i) | name(hi downto lo) -> (Sig, hi, lo)"""
        m = re.match(r"(\w+) ?(?:\((.*)\))?$", text.strip())
        if m.group(1) in self.alias:  # a tap of a delay line
            s = self.sigs[self.alias[m.group(1)]]
            return s, s.hi, s.lo
        s = self.sigs[m.group(1)]
        if m.group(2) is None:
            return s, s.hi, s.lo
        if " downto " in m.group(2):
            hi, lo = m.group(2).split(" downto ")
            return s, _int(hi, env), _int(lo, env)
        i = _int(m.group(2), env)
        return s, i, i

    def _value(self, text, env, want_w=None):
        """-> (value, width) of an expression on the right of <= or in a port map"""
        text = text.strip()
        parts = _split_top(text, "&")
        if len(parts) > 1:  # concatenation, left part on top
            v = w = 0
            for part in parts:
                pv, pw = self._value(part, env)
                v, w = (v << pw) | pv, w + pw
            return v, w
        parts = _split_kw(text, " and ")
        if len(parts) > 1:
            v, w = self._value(parts[0], env, want_w)
            for x in parts[1:]:
                v &= self._value(x, env, want_w)[0]
            return v, w
        parts = _split_top(text, "*")
        if len(parts) == 2 and all(x.startswith("unsigned") for x in parts):  # unsigned(a) * unsigned(b): the full product
            (a, wa), (b, wb) = (self._value(re.match(r"unsigned ?\((.*)\)$", x).group(1), env) for x in parts)
            return a * b, wa + wb
        m = re.match(r"(\w+) ?\( ?conv_integer ?\( ?unsigned ?\((\w+)\) ?\) ?\)$", text)
        if m and isinstance(env.get(m.group(1)), dict):  # a ROM built by a function, read at an index taken from a signal
            return env[m.group(1)][self._value(m.group(2), env)[0]], want_w
        m = re.match(r"(\w+) ?\( ?conv_integer ?\( ?(\w+) ?\) ?\)$", text)
        if m and m.group(1) in self.arrays:  # a memory read
            w, mem = self.arrays[m.group(1)]
            return mem.get(self._value(m.group(2), env)[0], 0), w
        if text.startswith('x"'):
            return int(text[2:-1], 16), 4 * (len(text) - 3)
        parts = _split_top(text, "+")
        if len(parts) == 2:  # x + '1' | x + 1: wraps at the width of x (std_logic_unsigned / signed)
            v, w = self._value(parts[0], env, want_w)
            return (v + int(parts[1].strip("'"))) & ((1 << w) - 1), w
        m = re.match(r"not ?\(([^()]*(?:\([^()]*\)[^()]*)*)\)$", text) or re.match(r"not (.+)$", text)
        if m:
            v, w = self._value(m.group(1), env, want_w)
            return (~v) & ((1 << w) - 1), w
        m = re.match(r"\( ?others ?=> ?(.*)\)$", text)
        if m:
            inner = m.group(1).strip()
            bit = int(inner.strip("'")) if inner.startswith("'") else self._value(inner, env)[0]
            assert want_w is not None
            return ((1 << want_w) - 1 if bit else 0), want_w
        m = re.match(r"\((.*others.*)\)$", text)
        if m:  # aggregate with positions: (0 => '1', 1 => '1', others => '0')
            v = 0
            for it in _split_top(m.group(1), ","):
                k, b = [x.strip() for x in it.split("=>")]
                if k != "others" and b == "'1'":
                    v |= 1 << int(k)
            return v, want_w
        m = re.match(r"sxt ?\((.*), ?([^,]+)\)$", text)
        if m:
            v, w = self._value(m.group(1), env)
            n = _int(m.group(2), env)
            return tw.sxt(v, w, n), n
        if text.startswith('"'):
            return int(text.strip('"'), 2), len(text) - 2
        if text.startswith("'"):
            return int(text.strip("'")), 1
        s, hi, lo = self._ref(text, env)
        return s.get(hi, lo), hi - lo + 1

    def run(self, inputs: dict) -> dict:
        for nm, s in self.sigs.items():
            if nm not in self.consts:
                s.val = s.known = 0
        for k, v in inputs.items():
            s = self.sigs[k]
            s.put(s.hi, s.lo, v)
        todo = list(self.pending)
        while todo:
            left = []
            for it in todo:
                try:
                    if it[0] == "assign":
                        _, lhs, rhs, env = it[:4]
                        s, hi, lo = self._ref(lhs, env)
                        if s.get_known(hi, lo):
                            continue  # driven from outside (an override of this evaluation)
                        v, _ = self._value(rhs, env, hi - lo + 1)
                        s.put(hi, lo, v)
                    elif it[0] == "proc":
                        acts = []
                        self._seq(it[1], it[2], acts)
                        for s, hi, lo, v in acts:  # all or nothing: a process that is not ready yet left no trace
                            if not s.get_known(hi, lo):
                                s.put(hi, lo, v)
                    else:
                        self._instance(id(it), *it[1:])
                except NotReady:
                    left.append(it)
            if len(left) == len(todo):
                break  # what is left waits for clk / rst style inputs or is undriven: fine as long as the outputs are known
            todo = left
        out = {}
        for p, (d, _) in self.ent.ports.items():
            if d == "out":
                s = self.sigs[p]
                if not s.full():
                    if p == "do_vl":
                        continue  # the valid strobe is timing
                    raise NotReady(p)
                out[p] = s.get(s.hi, s.lo)
        return out

    def _bool(self, cond, env) -> bool:
        """(a = '1') | (a(i) = '0') | c1 and c2 | c1 or c2, parenthesised at will"""
        cond = cond.strip()
        while cond.startswith("(") and _split_top(cond, "#") == [cond] and _balanced(cond[1:-1]) and cond.endswith(")"):
            cond = cond[1:-1].strip()
        for op_, fn in ((" or ", any), (" and ", all)):
            parts = _split_kw(cond, op_)
            if len(parts) > 1:
                return fn(self._bool(x, env) for x in parts)
        m = re.match(r"(.*?) ?= ?'([01])'$", cond)
        return self._value(m.group(1), env)[0] == int(m.group(2))

    def _seq(self, seq, env, acts):
        for node in seq:
            if node[0] == "assign":
                s, hi, lo = self._ref(node[1], env)
                v, _ = self._value(node[2], env, hi - lo + 1)
                acts.append((s, hi, lo, v))
            else:
                for cond, body in node[1]:
                    if self._bool(cond, env):
                        self._seq(body, env, acts)
                        break
                else:
                    self._seq(node[2], env, acts)

    def _instance(self, key, label, unit, gmap, pmap, env):
        if unit in ("dsp48e1", "dsp48e2"):
            series = "E1" if unit.endswith("1") else "E2"

            def val(port, w, dflt=0):
                a = pmap.get(port)
                if a is None or a == "open":
                    return dflt
                return self._value(a, env, w)[0]
            kw = dict(opmode=format(val("opmode", 7 if series == "E1" else 9), "0%db" % (7 if series == "E1" else 9)),
                      alumode=format(val("alumode", 4), "04b"), use_mult=gmap.get("use_mult", '"multiply"').strip('"').upper(),
                      a=val("a", 30), b=val("b", 18), c=val("c", 48), pcin=val("pcin", 48), carryin=val("carryin", 1),
                      carryinsel=format(val("carryinsel", 3), "03b"), carrycascin=val("carrycascin", 1),
                      use_simd=gmap.get("use_simd", '"one48"').strip('"').upper())
            assert val("inmode", 5) == 0 and val("d", 27 if series == "E2" else 25) == 0, "pre-adder / INMODE paths are not modelled"
            p, pcout, cy = tw.dsp48(series, **kw)
            for port, v, w in (("p", p, 48), ("pcout", pcout, 48), ("carrycascout", cy, 1)):
                a = pmap.get(port)
                if a and a != "open":
                    s, hi, lo = self._ref(a, env)
                    assert hi - lo + 1 == w
                    s.put(hi, lo, v)
            return
        sub_ent = entity(unit)
        sub = self.subs.get(key)
        if sub is None:
            g = {}
            for k, v in gmap.items():
                g[k] = v.strip('"') if v.startswith('"') else (_int(v, env) if not (v in env and isinstance(env[v], str)) else env[v])
            sub = self.subs[key] = Inst(sub_ent, g)
        ins = {}
        for port, (d, _) in sub_ent.ports.items():
            if d == "in" and port in pmap and port not in ("clk", "rst"):
                ins[port] = self._value(pmap[port], env, sub.sigs[port].width)[0]
        outs = sub.run(ins)
        for port, v in outs.items():
            a = pmap.get(port)
            if a and a != "open":
                s, hi, lo = self._ref(a, env)
                s.put(hi, lo, v)


class CycleSim(Inst):
    """A clocked evaluation of ONE entity that is all registers, counters and memories (the delay lines): every signal starts at 0,
    unregistered concurrent assignments are combinational, `x <= y when rising_edge(clk)` and the clocked processes take their inputs
    before the edge and show their outputs after it (signal assignment semantics: the last assignment in a process wins)."""

    def __init__(self, ent, generics):
        super().__init__(ent, generics)
        for s in self.sigs.values():
            s.val, s.known = 0, (1 << s.width) - 1

    def _comb(self):
        for _ in range(8):  # a few passes settle the short combinational chains of these entities
            changed = False
            for it in self.pending:
                if it[0] == "assign" and not it[4]:
                    s, hi, lo = self._ref(it[1], it[3])
                    v, _ = self._value(it[2], it[3], hi - lo + 1)
                    if s.get(hi, lo) != v & ((1 << (hi - lo + 1)) - 1):
                        self._force(s, hi, lo, v)
                        changed = True
            if not changed:
                return

    @staticmethod
    def _force(s, hi, lo, v):
        if hi < lo:
            return
        w = hi - lo + 1
        m = ((1 << w) - 1) << (lo - s.lo)
        s.val = (s.val & ~m) | ((v & ((1 << w) - 1)) << (lo - s.lo))

    def _seq_clocked(self, seq, env, acts, writes):
        for node in seq:
            if node[0] == "assign":
                m = re.match(r"(\w+) ?\( ?conv_integer ?\( ?(\w+) ?\) ?\)$", node[1])
                if m and m.group(1) in self.arrays:
                    writes.append((m.group(1), self._value(m.group(2), env)[0], self._value(node[2], env, self.arrays[m.group(1)][0])[0]))
                    continue
                s, hi, lo = self._ref(node[1], env)
                acts.append((s, hi, lo, self._value(node[2], env, hi - lo + 1)[0]))
            else:
                for cond, body in node[1]:
                    if self._bool(cond, env):
                        self._seq_clocked(body, env, acts, writes)
                        break
                else:
                    self._seq_clocked(node[2], env, acts, writes)

    def clock(self, inputs: dict) -> dict:
        """drive the inputs, settle, take one rising edge, settle; -> the output ports after the edge"""
        for k, v in inputs.items():
            s = self.sigs[k]
            self._force(s, s.hi, s.lo, v)
        self._comb()
        acts, writes = [], []
        for it in self.pending:
            if it[0] == "assign" and it[4]:
                if it[5] is not None and not self._bool(it[5], it[3]):
                    continue  # clock enable low: the register keeps its value
                s, hi, lo = self._ref(it[1], it[3])
                acts.append((s, hi, lo, self._value(it[2], it[3], hi - lo + 1)[0]))
            elif it[0] == "proc":
                self._seq_clocked(it[1], it[2], acts, writes)
        for s, hi, lo, v in acts:
            self._force(s, hi, lo, v)
        for name, idx, v in writes:
            self.arrays[name][1][idx] = v
        self._comb()
        return {p: self.sigs[p].val for p, (d, _) in self.ent.ports.items() if d == "out"}


def run_delay_line(unit: str, nfft: int, stage: int, frames_a, frames_b, gap: int = 0):
    """int_delay_line / int_delay_wrap clocked from the text: frames of N/2 beats (two lanes of words) in, with `gap` idle clocks between
    frames -> the beats that come out with DO_VL = '1', as two lanes"""
    sim = CycleSim(entity(unit), {"nfft": nfft, "stage": stage, "nwidth": 32})
    sim.clock({"rst": 1, "di_en": 0, "di_aa": 0, "di_bb": 0})
    oa, ob = [], []

    def tick(a, b, en):
        o = sim.clock({"rst": 0, "di_en": en, "di_aa": a, "di_bb": b})
        if o["do_vl"]:
            oa.append(o["do_aa"])
            ob.append(o["do_bb"])

    for fa, fb in zip(frames_a, frames_b):
        for a, b in zip(fa, fb):
            tick(a, b, 1)
        for _ in range(gap):
            tick(0, 0, 0)
    for _ in range(4 << nfft):
        tick(0, 0, 0)
    return oa, ob


_ENT = {}


def forget():
    """drop every parsed / elaborated entity (the tests mutate the text that _load returns)"""
    _ENT.clear()
    _INST.clear()


def entity(name: str) -> Entity:
    if name not in _ENT:
        _ENT[name] = Entity(name)
    return _ENT[name]


_INST = {}


def evaluate(name: str, generics: dict, inputs: dict) -> dict:
    """generics / inputs by lower-case name; string generics lower case ("new" / "old", "add" / "sub"); values are raw vectors"""
    key = (name, tuple(sorted(generics.items())))
    if key not in _INST:
        _INST[key] = Inst(entity(name), generics)
    return _INST[key].run(inputs)


# ------------------------------------------------------------------------------------------------ comparisons with the twin

def operand(rng, w):
    lo, hi = -(1 << (w - 1)), (1 << (w - 1)) - 1
    k = rng.random()
    if k < 0.6:
        return rng.randint(lo, hi)
    if k < 0.75:
        return rng.choice((lo, hi, -1, 0, 1))
    v = rng.randint(lo, hi)
    cut = rng.choice((17, 18, 34, 48))
    if cut < w:
        m = (1 << cut) - 1
        v = (v | m) if rng.random() < 0.5 else (v & ~m)
    return max(lo, min(hi, v))


MULTS = {"mlt42x18_dsp48e1": (42, 18), "mlt44x18_dsp48e2": (44, 18), "mlt35x25_dsp48e1": (35, 25), "mlt35x27_dsp48e2": (35, 27),
         "mlt59x18_dsp48e1": (59, 18), "mlt61x18_dsp48e2": (61, 18), "mlt52x25_dsp48e1": (52, 25), "mlt52x27_dsp48e2": (52, 27)}


def check_mult(name, n, rng):
    aw, bw = MULTS[name]
    bad = 0
    for _ in range(n):
        a, b = operand(rng, aw), operand(rng, bw)
        got = evaluate(name, {}, {"mlt_a": tw.vec(a, aw), "mlt_b": tw.vec(b, bw)})["mlt_p"]
        want = getattr(tw, name)(tw.vec(a, aw), tw.vec(b, bw))
        if got != want or tw.signed(got, aw + bw) != a * b:
            bad += 1
            print("MISMATCH", name, a, b, got, want)
    return bad


def check_cmult(dtw, twd, xser, n, rng):
    """int_cmult_dsp48 elaborated from the reference text against the hand-wired twin and oracle_py"""
    from oracle import oracle_py as op
    bad = 0
    for _ in range(n):
        v = [operand(rng, dtw), operand(rng, dtw), operand(rng, twd), operand(rng, twd)]
        r = evaluate("int_cmult_dsp48", {"dtw": dtw, "twd": twd, "xser": xser.lower()},
                     {"di_re": tw.vec(v[0], dtw), "di_im": tw.vec(v[1], dtw), "ww_re": tw.vec(v[2], twd), "ww_im": tw.vec(v[3], twd)})
        got = (r["do_re"], r["do_im"])
        want = tw.int_cmult_dsp48(tw.vec(v[0], dtw), tw.vec(v[1], dtw), tw.vec(v[2], twd), tw.vec(v[3], twd), dtw, twd, xser)
        ref = op.cmult(v[0], v[1], v[2], v[3], dtw, twd, xser == "NEW")
        if got != want or (tw.signed(got[0], dtw), tw.signed(got[1], dtw)) != ref:
            bad += 1
            print("MISMATCH cmult", dtw, twd, xser, v, got, want, ref)
    return bad


def check_addsub(dspw, xser, n, rng):
    bad = 0
    for _ in range(n):
        v = [operand(rng, dspw) for _ in range(4)]
        r = evaluate("int_addsub_dsp48", {"dspw": dspw, "xser": xser.lower()},
                     {"ia_re": tw.vec(v[0], dspw), "ia_im": tw.vec(v[1], dspw), "ib_re": tw.vec(v[2], dspw), "ib_im": tw.vec(v[3], dspw)})
        got = (r["ox_re"], r["ox_im"], r["oy_re"], r["oy_im"])
        want = tw.int_addsub_dsp48(*(tw.vec(x, dspw) for x in v), dspw, xser)
        exact = (v[0] + v[2], v[1] + v[3], v[0] - v[2], v[1] - v[3])
        if got != tuple(want) or tuple(tw.signed(g, dspw + 1) for g in got) != exact:
            bad += 1
            print("MISMATCH addsub", dspw, xser, v, got, want)
    return bad


def check_fly(kind, dtw, tfw, scale, rnd, stage, odd, xser, n, rng):
    """int_dif2_fly / int_dit2_fly elaborated from the text (its adder, rounding and negation processes, its multiplier) against the twin
    and oracle_py.  dt_sw -- the STAGE 1 toggle that pr_cnt flips on every valid beat -- is driven from outside."""
    from oracle import oracle_py as op
    name = "int_dif2_fly" if kind == "dif" else "int_dit2_fly"
    g = {"stage": stage, "scale": scale, "dtw": dtw, "tfw": tfw, "rndmode": rnd, "xser": xser.lower()}
    wo = dtw - scale + 1
    bad = 0
    for _ in range(n):
        v = [operand(rng, dtw) for _ in range(4)]
        ww = (operand(rng, tfw), operand(rng, tfw))
        ins = {"ia_re": tw.vec(v[0], dtw), "ia_im": tw.vec(v[1], dtw), "ib_re": tw.vec(v[2], dtw), "ib_im": tw.vec(v[3], dtw),
               "ww_re": tw.vec(ww[0], tfw), "ww_im": tw.vec(ww[1], tfw), "in_en": 1, "rst": 0}
        if stage == 1:
            ins["dt_sw"] = odd
        r = evaluate(name, g, ins)
        got = (r["oa_re"], r["oa_im"], r["ob_re"], r["ob_im"])
        f = tw.int_dif2_fly if kind == "dif" else tw.int_dit2_fly
        want = f(*(tw.vec(x, dtw) for x in v), tw.vec(ww[0], tfw), tw.vec(ww[1], tfw), stage=stage, scale=scale, dtw=dtw, tfw=tfw, rndmode=rnd,
                 xser=xser, dt_sw=odd)
        o = (op.dif_fly if kind == "dif" else op.dit_fly)((v[0], v[1]), (v[2], v[3]), ww, stage, dtw, tfw, scale, rnd, odd, xser == "NEW")
        ref = tuple(op.sgn(x, wo) for x in (o[0][0], o[0][1], o[1][0], o[1][1]))
        if got != tuple(want) or tuple(tw.signed(x, wo) for x in got) != ref:
            bad += 1
            print("MISMATCH fly", kind, g, odd, v, ww, got, want, ref)
    return bad


def stage_schedule(top: str, generics: dict):
    """The stage wiring of int_fftNk / int_ifftNk as its generate loops read: for every instance the unit and its evaluated generic map
    (nothing is evaluated: the delay lines are clocked memories).  -> [(label, unit, {generic: value}, ii)]"""
    inst = Inst(entity(top), generics)
    out = []
    for it in inst.pending:
        if it[0] != "inst":
            continue
        _, label, unit, gmap, _pmap, env = it
        g = {}
        for k, v in gmap.items():
            g[k] = v.strip('"') if v.startswith('"') else (env[v] if (v in env and not isinstance(env[v], int)) else _int(v, env))
        out.append((label, unit, g, env.get("ii")))
    return out


def delay_block_log2(unit: str, nfft: int, stage: int) -> int:
    """N_INV of int_delay_line / int_delay_wrap as its constant declaration reads: the cross-commutation works on blocks of 2^N_INV words"""
    t = _load(unit)
    m = re.search(r"constant n_inv ?: ?integer ?:= ?([^;]+);", t)
    return _int(m.group(1), {"nfft": nfft, "stage": stage})


def check_delay(unit, nfft, stage, gap, rng, frames=4):
    """int_delay_line / int_delay_wrap clocked from the text on back-to-back frames (int_delay_wrap also with idle clocks between frames)
    against the cross-commutation the oracle's stream form uses (oracle_py._rev2rdx = fn_rev2rdx of math/fn_radix2.m).  int_delay_wrap
    hands the tail of a frame out while the next one comes in, so its last block stays inside: a prefix is compared."""
    from oracle import oracle_py as op
    half = 1 << (nfft - 1)
    fa = [[rng.randrange(1 << 31) for _ in range(half)] for _ in range(frames)]
    fb = [[rng.randrange(1 << 31) for _ in range(half)] for _ in range(frames)]
    oa, ob = run_delay_line(unit, nfft, stage, fa, fb, gap)
    ea, eb = [], []
    for a, b in zip(fa, fb):
        x, y = op._rev2rdx(a, b, 1 << delay_block_log2(unit, nfft, stage))
        ea += x
        eb += y
    ok = oa == ea[:len(oa)] and ob == eb[:len(ob)] and len(oa) >= (frames - 1) * half and (unit != "int_delay_line" or len(oa) == len(ea))
    if not ok:
        print("MISMATCH delay", unit, nfft, stage, gap, len(oa), len(ea))
    return 0 if ok else 1


def check_twiddles(stage, awd, xser, use_mlt, n, rng):
    """rom_twiddle_int elaborated from the text -- the ROM its function fills from MATH_PI / COS / SIN, the quadrant rotation process, the
    address slicing, the Taylor sub-entity for STAGE >= 11 -- read at counter value cnt, against oracle_py.twiddles(stage)[cnt]"""
    from oracle import oracle_py as op
    want = op.twiddles(stage, awd, xser == "NEW")
    bad = 0
    for _ in range(n):
        cnt = rng.randrange(1 << stage)
        r = evaluate("rom_twiddle_int", {"awd": awd, "nfft": 20, "stage": stage, "use_mlt": use_mlt, "xser": xser.lower()},
                     {"cnt": cnt, "rst": 0, "ww_en": 1})
        got = (tw.signed(r["ww_re"], awd), tw.signed(r["ww_im"], awd))
        if got != want[cnt]:
            bad += 1
            print("MISMATCH twiddle", stage, awd, xser, use_mlt, cnt, got, want[cnt])
    return bad


def check_taylor(awd, ii, xser, use_mlt, n, rng):
    """row_twiddle_tay elaborated from the text (both forms of MATHPI * cnt: the ROM built by read_rom and the multiplier process) fed
    the way rom_twiddle_int feeds it, against the twin and -- through the twin's own test -- oracle_py.twiddles"""
    from oracle import oracle_py as op
    rom = op._rom(9, awd)
    bad = 0
    for _ in range(n):
        re_, im_ = rom[rng.randrange(512)]
        if rng.random() < 0.5:
            re_, im_ = im_, -re_
        ww = tw.vec(re_, awd) | (tw.vec(im_, awd) << awd)
        cnt = rng.randrange(1 << (ii + 1))
        r = evaluate("row_twiddle_tay", {"awd": awd, "xser": xser.lower(), "use_mlt": use_mlt, "ii": ii}, {"rom_ww": ww, "rom_cnt": cnt, "rstp": 0})
        want = tw.row_twiddle_tay(ww, cnt, awd, xser, ii, use_mlt)
        if (r["rom_re"], r["rom_im"]) != want:
            bad += 1
            print("MISMATCH taylor", awd, ii, xser, use_mlt, cnt, r, want)
    return bad


def main():
    if not available():
        print("reference not present: nothing to do")
        return 0
    from oracle import oracle_py as op
    n = int(sys.argv[sys.argv.index("--per-case") + 1]) if "--per-case" in sys.argv else 40
    rng = random.Random(int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 2026)
    bad = cases = sets = 0
    for name in MULTS:
        bad += check_mult(name, 10 * n, rng)
        cases += 1
        sets += 10 * n
    by = {}
    for new in (True, False):
        for t in range(8, 28):
            for w in range(8, 79):
                r = op.cmult_regime(w, t, new)
                if r:
                    by.setdefault((r, new), []).append((w, t))
    for (r, new), lst in sorted(by.items()):
        pick = rng.sample(lst, min(30, len(lst))) + [min(lst), max(lst)]
        b = sum(check_cmult(w, t, "NEW" if new else "OLD", n, rng) for w, t in pick)
        print("%-8s %s: %3d width pairs x %d operand sets, %d mismatches" % (r, "NEW" if new else "OLD", len(pick), n, b), flush=True)
        bad += b
        cases += len(pick)
        sets += len(pick) * n
    for xser in ("NEW", "OLD"):
        b = sum(check_addsub(d, xser, n, rng) for d in range(4, 96))
        print("addsub   %s: DSPW 4 .. 95 x %d operand sets, %d mismatches" % (xser, n, b), flush=True)
        bad += b
        cases += 92
        sets += 92 * n
    from oracle import oracle_py as op2
    for kind in ("dif", "dit"):
        b = k = 0
        for _ in range(60):
            xser = rng.choice(("NEW", "OLD"))
            dtw, tfw = rng.randint(8, 50), rng.choice((12, 16, 18, 19, 24, 25))
            scale = rng.randint(0, 1)
            rnd = rng.randint(0, 1) if scale else 0
            stage, odd = rng.choice((0, 1, 1, 2, 5, 12)), rng.randint(0, 1)
            if stage > 1 and op2.cmult_regime(dtw + 1 - scale if kind == "dif" else dtw, tfw, xser == "NEW") is None:
                continue
            b += check_fly(kind, dtw, tfw, scale, rnd, stage, odd, xser, n, rng)
            k += 1
        print("int_%s2_fly: %d random (widths, mode, STAGE, XSER) x %d operand sets, %d mismatches" % (kind, k, n, b), flush=True)
        bad += b
        cases += k
        sets += k * n
    b = k = 0
    for xser in ("NEW", "OLD"):
        for use_mlt in (False, True):
            for awd in (12, 16, 18, 19, 24, 25):
                for ii in range(8):  # STAGE 11 .. 18; ii = 8 (N = 2^20) does not elaborate in the reference: this project's extension
                    b += check_taylor(awd, ii, xser, use_mlt, max(4, n // 4), rng)
                    k += 1
    print("row_twiddle_tay: %d (AWD, ii, XSER, USE_MLT) x %d operand sets, %d mismatches" % (k, max(4, n // 4), b), flush=True)
    bad += b
    cases += k
    sets += k * max(4, n // 4)
    b = k = 0
    for xser in ("NEW", "OLD"):
        for awd in (12, 16, 18, 19, 24, 25):
            for stage in range(2, 19):
                b += check_twiddles(stage, awd, xser, bool(stage & 1) and stage >= 11, max(4, n // 4), rng)
                k += 1
    print("rom_twiddle_int: %d (STAGE 2 .. 18, AWD, XSER) x %d counter values, %d mismatches against oracle_py.twiddles" % (k, max(4, n // 4), b),
          flush=True)
    bad += b
    cases += k
    sets += k * max(4, n // 4)
    b = k = 0
    for nfft in range(3, 9):
        for stage in range(nfft - 1):
            for unit, gap in (("int_delay_line", 0), ("int_delay_wrap", 0), ("int_delay_wrap", 3)):
                b += check_delay(unit, nfft, stage, gap, rng)
                k += 1
    print("delay lines: %d (entity, NFFT 3 .. 8, STAGE, gap) clocked from the text, 4 frames each, %d mismatches against the oracle's commutation"
          % (k, b), flush=True)
    bad += b
    cases += k
    print("rtl_interp: %d elaborations of the reference's own text, %d operand sets, %d mismatches against the hand-wired twin / oracle_py"
          % (cases, sets, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
