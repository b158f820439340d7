#!/usr/bin/env python3
"""A cycle-based simulation of the reference's int_fftNk / int_ifftNk FROM ITS OWN VHDL TEXT -- TEST INFRASTRUCTURE, this container only.

    python tools/rtl_sim.py [--nfft 4] [--frames 3] | --sweep | --kit [case ...] | --fuzz COUNT SEED | --fuzz-wrappers COUNT SEED  (needs /root/reference; reads it, stores nothing)

tools/rtl_interp.py evaluates the arithmetic entities of the reference as dataflow networks.  This tool goes the rest of the way: it
elaborates a whole core -- the generate loops of int_fftNk.vhd, every butterfly, twiddle generator, aligner and delay line under it, down to
the DSP48 primitives -- as a hierarchy of clocked nodes and runs it cycle by cycle on frames of input beats:

  * every signal is a register or a wire of the text: unregistered concurrent assignments are combinational, `x <= y when rising_edge(clk)`
    and clocked processes sample before the edge and show after it (two-phase: compute all next values, then commit), memories included;
  * arrays of vectors (the per-stage buses of int_fftNk, the delay chains `z <= z(z'left-1 downto 0) & x`) are arrays of signals;
  * a DSP48E1 / DSP48E2 instance is the slice model of oracle/dsp48_twin.py BEHIND THE PIPELINE REGISTERS ITS GENERIC MAP ASKS FOR
    (AREG / BREG 0, 1, 2; CREG; MREG; PREG; PCIN = the neighbour's registered P; CARRYCASCIN = its registered carry): the latencies that the
    aligners, the valid strobes and the twiddle counters are built around come from the text's own generics (UG479 / UG579 for what a
    register stage is).
  * the wrappers of src/vhdl/main run the same way with their I/O buffers (inbuf_half_path, outbuf_half_path, int_bitrev_order,
    iobuf_flow_int2): memories behind shared variables, integer signals, and the buffers' own functions (bit_pair; str_array / hi_bits, which
    build the address-increment tables) executed from the text (class VecFn).
The DO_VAL-qualified output beats are compared with oracle_py on the same frames.  If the pipeline of the text did not line up with its own
strobes, or the oracle misread any of it, the frames would differ.

Not a VHDL simulator: no delta cycles (one combinational settle per phase), no 'U' / 'X' (everything starts at 0), std_logic as bits, only
the constructs these files use.  Parity stays unpinned by the rules of this build (the slice model is a stand-in for unisim) -- but this is
the reference's own text, end to end, producing the oracle's numbers.
"""
from __future__ import annotations

import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rtl_interp as R  # noqa: E402
from oracle import dsp48_twin as tw  # noqa: E402

available = R.available


class Wire:
    """A std_logic_vector (or std_logic) signal: value now, value after the next edge."""
    __slots__ = ("hi", "lo", "val")

    def __init__(self, hi, lo):
        self.hi, self.lo, self.val = hi, lo, 0

    @property
    def width(self):
        return max(0, self.hi - self.lo + 1)

    def get(self, hi, lo):
        if hi < lo:
            return 0
        assert self.lo <= lo <= hi <= self.hi, "slice (%d downto %d) outside (%d downto %d)" % (hi, lo, self.hi, self.lo)
        return (self.val >> (lo - self.lo)) & ((1 << (hi - lo + 1)) - 1)

    def set(self, hi, lo, v):
        if hi < lo:
            return False
        assert self.lo <= lo <= hi <= self.hi, "slice (%d downto %d) outside (%d downto %d)" % (hi, lo, self.hi, self.lo)
        w = hi - lo + 1
        m = ((1 << w) - 1) << (lo - self.lo)
        new = (self.val & ~m) | ((v & ((1 << w) - 1)) << (lo - self.lo))
        ch = new != self.val
        self.val = new
        return ch


class Dsp:
    """One DSP48 slice behind its pipeline registers."""

    def __init__(self, series, gmap, pmap, env, parent):
        self.series, self.pmap, self.env, self.parent = series, pmap, env, parent
        gi = lambda k, d: int(gmap.get(k, str(d)))  # noqa: E731
        self.areg, self.breg, self.creg, self.mreg, self.preg = gi("areg", 1), gi("breg", 1), gi("creg", 1), gi("mreg", 1), gi("preg", 1)
        self.use_mult = gmap.get("use_mult", '"multiply"').strip('"').upper()
        self.use_simd = gmap.get("use_simd", '"one48"').strip('"').upper()
        assert self.preg == 1, "every slice of the reference registers P"
        self.a = [0, 0]  # A1, A2
        self.b = [0, 0]
        self.c = 0
        self.m = None    # registered (a, b) operands of the multiplier stage
        self.p = 0
        self.cy = 0
        self.nxt = None

    def _in(self, port, w, dflt=0):
        a = self.pmap.get(port)
        if a is None or a == "open":
            return dflt
        return self.parent.value(a, self.env, w)[0]

    def _ab(self, regs, n, direct):
        return direct if n == 0 else regs[1]

    def compute(self):
        """next state from the present one (phase 1)"""
        ws = 7 if self.series == "E1" else 9
        a_in, b_in, c_in = self._in("a", 30), self._in("b", 18), self._in("c", 48)
        rst = self._in({"E1": "rsta", "E2": "rsta"}[self.series], 1)
        a_eff = a_in if self.areg == 0 else self.a[1]
        b_eff = b_in if self.breg == 0 else self.b[1]
        c_eff = c_in if self.creg == 0 else self.c
        if self.use_mult == "MULTIPLY" and self.mreg:
            ma, mb = self.m if self.m is not None else (0, 0)
        else:
            ma, mb = a_eff, b_eff
        p, _, cy = tw.dsp48(self.series, opmode=format(self._in("opmode", ws), "0%db" % ws), alumode=format(self._in("alumode", 4), "04b"),
                            use_mult=self.use_mult, a=ma if self.use_mult == "MULTIPLY" else a_eff, b=mb if self.use_mult == "MULTIPLY" else b_eff,
                            c=c_eff, pcin=self._in("pcin", 48), carryin=self._in("carryin", 1), carryinsel=format(self._in("carryinsel", 3), "03b"),
                            carrycascin=self._in("carrycascin", 1), use_simd=self.use_simd)
        na = [a_in, self.a[0]] if self.areg == 2 else [a_in, a_in]
        nb = [b_in, self.b[0]] if self.breg == 2 else [b_in, b_in]
        if rst:
            self.nxt = ([0, 0], [0, 0], 0, None, 0, 0)
        else:
            self.nxt = (na, nb, c_in, (a_eff, b_eff), p, cy)

    def commit(self):
        na, nb, c, m, p, cy = self.nxt
        if self.areg == 2:
            self.a = [na[0], na[1]]
        else:
            self.a = [na[0], na[0]]
        if self.breg == 2:
            self.b = [nb[0], nb[1]]
        else:
            self.b = [nb[0], nb[0]]
        self.c, self.m, self.p, self.cy = c, m, p, cy

    def drive(self):
        """registered outputs onto the parent's signals; -> changed?"""
        ch = False
        for port, v, w in (("p", self.p, 48), ("pcout", self.p, 48), ("carrycascout", self.cy, 1)):
            a = self.pmap.get(port)
            if a and a != "open":
                ch |= self.parent.assign(a, self.env, v)
        return ch


class VecFn:
    """The functions of buffers/iobuf_flow_int2.vhd (str_array, hi_bits) work on std_logic_vector variables and return arrays: executed from
    the text on a small value model -- integers / booleans as Python values, a vector variable as [value, width], an array as {index: value}."""

    def __init__(self, entity_name):
        self.fns = {}
        for m in R.FUNC.finditer(R._load(entity_name)):
            params = [nm.strip() for x in m.group(2).split(";") for nm in x.split(":")[0].split(",")] if m.group(2) else []
            self.fns[m.group(1)] = (params, m.group(3), R._parse_fn(m.group(4)))

    def call(self, name, args, env):
        params, decls, ast = self.fns[name]
        scope = dict(env, true=True, false=False)
        scope.update(zip(params, args))
        width = {}   # vector variables: name -> width
        for st in R._split_top(decls, ";"):
            m = re.match(r"variable (\w+) ?: ?std_logic_vector ?\((.*) downto (.*)\)", st)
            if m:
                width[m.group(1)] = R._int(m.group(2), scope) - R._int(m.group(3), scope) + 1
                scope[m.group(1)] = 0
            else:
                m = re.match(r"variable (\w+) ?: ?\w+", st)
                if m:
                    scope[m.group(1)] = {}

        def attr(t):
            return re.sub(r"(\w+)'left", lambda k: str(width[k.group(1)] - 1), t)

        def val(t):
            t = attr(t.strip())
            m = re.match(r"\((.*others.*)\)$", t)
            if m:   # (k => '1', others => '0')
                v = 0
                for it in R._split_top(m.group(1), ","):
                    k, b = [x.strip() for x in it.split("=>")]
                    if k != "others" and b == "'1'":
                        v |= 1 << R._int(k, scope)
                return v
            m = re.match(r"(\w+) ?\((.*?)\) ?\((.*) downto (.*)\)$", t)
            if m and isinstance(scope.get(m.group(1)), dict):   # arr(j)(hi downto lo)
                hi, lo = R._int(m.group(3), scope), R._int(m.group(4), scope)
                return (scope[m.group(1)][R._int(m.group(2), scope)] >> lo) & ((1 << (hi - lo + 1)) - 1)
            m = re.match(r"(\w+) ?\((.*?)\) ?\(([^()]*)\)$", t)
            if m and isinstance(scope.get(m.group(1)), dict):   # arr(j)(bit)
                return (scope[m.group(1)][R._int(m.group(2), scope)] >> R._int(m.group(3), scope)) & 1
            m = re.match(r"(\w+) ?\((.*) downto (.*)\)$", t)
            if m and m.group(1) in width:   # a slice of a vector variable
                hi, lo = R._int(m.group(2), scope), R._int(m.group(3), scope)
                return (scope[m.group(1)] >> lo) & ((1 << (hi - lo + 1)) - 1)
            if re.match(r"\( ?others ?=> ?'0' ?\)$", t):
                return 0
            if t in scope:
                return scope[t]
            return R._int(t, scope)

        def run(seq):
            for node in seq:
                if node[0] == "set":
                    tgt, v = attr(node[1]), val(node[2])
                    m = re.match(r"(\w+) ?\((.*) downto (.*)\)$", tgt)
                    if m:
                        hi, lo = R._int(m.group(2), scope), R._int(m.group(3), scope)
                        mask = ((1 << (hi - lo + 1)) - 1) << lo
                        scope[m.group(1)] = (scope[m.group(1)] & ~mask) | ((v << lo) & mask)
                        continue
                    m = re.match(r"(\w+) ?\((.*)\)$", tgt)
                    if m and m.group(1) in width:      # one bit of a vector
                        i = R._int(m.group(2), scope)
                        scope[m.group(1)] = (scope[m.group(1)] & ~(1 << i)) | ((v & 1) << i)
                    elif m:                            # one element of an array
                        scope[m.group(1)][R._int(m.group(2), scope)] = v
                    else:
                        scope[tgt] = v
                elif node[0] == "if":
                    for c, body in node[1]:
                        if R._cond(c, scope):
                            r = run(body)
                            if r is not None:
                                return r
                            break
                    else:
                        r = run(node[2])
                        if r is not None:
                            return r
                elif node[0] == "for":
                    for v in range(R._int(node[2], scope), R._int(node[3], scope) + 1):
                        scope[node[1]] = v
                        r = run(node[4])
                        if r is not None:
                            return r
                else:
                    return ("ret", val(node[1]))
            return None

        return run(ast)[1]


class Node:
    """One elaborated entity instance."""

    def __init__(self, name, generics, parent=None, pmap=None, penv=None):
        self.ent = R.entity(name)
        self.parent, self.pmap, self.penv = parent, pmap or {}, penv or {}
        self.env = {}
        for g, dflt in self.ent.generics:
            self.env[g] = generics.get(g, dflt.strip('"') if dflt.startswith('"') else (int(dflt) if dflt.lstrip("-").isdigit() else
                                                                                          {"true": True, "false": False}.get(dflt)))
        self.w = {}         # name -> Wire
        self.arr = {}       # name -> (hi, lo, [Wire])  arrays of vectors, element index hi downto lo
        self.mem = {}       # name -> (word width, {index: value})
        self.types = {}     # array type name -> ("vec", hi_expr, lo_expr, elem_hi, elem_lo) evaluated
        self.funcs = dict(self.ent.functions)
        self.comb, self.regs, self.procs, self.kids = [], [], [], []
        self.consts = set()
        self._vc, self._rc, self._cc, self._keep = {}, {}, {}, {}   # parsed expressions / references / conditions, by (text, region)
        self._cw = self._ca = self._kc = None
        self.ints = set()   # integer signals (iobuf_flow_int2's in_cnt, wr_rng)
        self.vecfn = None
        self._decls(self.ent.decls, self.env)
        for p, (_, rng) in self.ent.ports.items():
            if rng is None:
                self.w[p] = Wire(0, 0)
            else:
                hi, lo = rng.split(" downto ")
                self.w[p] = Wire(R._int(hi, self.env), R._int(lo, self.env))
        self._region(self.ent.body, dict(self.env))

    # ---- elaboration ---------------------------------------------------------------------------------------------------------------
    def _decls(self, text, env):
        for st in R._split_top(text, ";"):
            if st.startswith(("signal ", "shared variable ")):
                st = re.sub(r" ?:= ?.*$", "", st)
            m = re.match(r"shared variable (\w+) ?: ?(\w+)$", st)
            if m:   # inbuf_half_path: a memory written with `:=` from one process (read before the write in the text: read-first)
                self.mem[m.group(1)] = (self.types[m.group(2)][2] - self.types[m.group(2)][3] + 1, {})
                continue
            m = re.match(r"shared variable (\w+) ?: ?(\w+) ?\((.*) downto (.*)\)$", st)
            if m:   # iobuf_flow_int2: an unconstrained array type, constrained at the variable
                self.mem[m.group(1)] = (self.types[m.group(2)][2] - self.types[m.group(2)][3] + 1, {})
                continue
            m = re.match(r"signal ([\w, ]+) ?: ?integer range .*$", st)
            if m:
                for nm in m.group(1).split(","):
                    self.w[nm.strip()] = Wire(31, 0)
                    self.ints.add(nm.strip())
                continue
            m = re.match(r"type (\w+) is array ?\(.*\) of integer$", st)
            if m:
                continue
            m = re.match(r"constant (\w+) ?: ?(\w+)(?: ?\((.*?) downto (.*?)\))? ?:= ?(.*)$", st)
            if m:
                name, init = m.group(1), m.group(5).strip()
                k = re.match(r"(\w+)(?: ?\((.*)\))?$", init)
                try:
                    if k and k.group(1) in self.funcs and self.vecfn is None:
                        self.vecfn = VecFn(self.ent.name)
                    if k and k.group(1) in self.funcs and "std_logic_vector" in self.vecfn.fns[k.group(1)][1]:
                        args = [env[a.strip()] if a.strip() in env else {"true": True, "false": False}[a.strip()] if a.strip() in ("true", "false")
                                else R._int(a, env) for a in R._split_top(k.group(2), ",")]
                        env[name] = self.vecfn.call(k.group(1), args, env)   # an array constant: {index: value}
                        self.env.setdefault(name, env[name])
                        continue
                    if k and k.group(1) in self.funcs:
                        args = [env[a.strip()] if a.strip() in env else R._int(a, env) for a in R._split_top(k.group(2), ",")] if k.group(2) else []
                        env[name] = R.call_function(self.funcs, k.group(1), args, env)
                    elif m.group(3):
                        kk = re.match(r"std_logic_vector ?\( ?conv_unsigned ?\((.*), ?(\d+) ?\) ?\)$", init)
                        self.w[name] = Wire(R._int(m.group(3), env), R._int(m.group(4), env))
                        self.w[name].val = R._int(kk.group(1), env) & ((1 << self.w[name].width) - 1)
                        self.consts.add(name)
                        continue
                    else:  # an expression, possibly calling the entity's own functions: addsub_delay(dtw+scale+rndmode)+rndmode
                        scope = dict(env)
                        for fn in self.funcs:
                            scope[fn] = (lambda f: (lambda *a: R.call_function(self.funcs, f, list(a), env)))(fn)
                        env[name] = int(R._fn_eval(init, scope))
                    self.env.setdefault(name, env[name])
                except Exception as exc:
                    raise AssertionError("constant %s of %s: %r" % (name, self.ent.name, exc))
                continue
            m = re.match(r"type (\w+) is array ?\( ?integer range <> ?\) of std_logic_vector ?\((.*) downto (.*)\)$", st)
            if m:
                self.types[m.group(1)] = (0, 0, R._int(m.group(2), env), R._int(m.group(3), env), "to")
                continue
            m = re.match(r"type (\w+) is array ?\((.*?) (downto|to) (.*?)\) of std_logic_vector ?\((.*) downto (.*)\)$", st)
            if m:
                a, b = R._int(m.group(2), env), R._int(m.group(4), env)
                self.types[m.group(1)] = (max(a, b), min(a, b), R._int(m.group(5), env), R._int(m.group(6), env), m.group(3))
                continue
            m = re.match(r"signal ([\w, ]+) ?: ?std_logic(?:_vector ?\((.*) downto (.*)\))?$", st)
            if m:
                rng = (R._int(m.group(2), env), R._int(m.group(3), env)) if m.group(2) else (0, 0)
                for nm in m.group(1).split(","):
                    self.w[nm.strip()] = Wire(*rng)
                continue
            m = re.match(r"signal ([\w, ]+) ?: ?(\w+)$", st)
            if m and m.group(2) in self.types:
                hi, lo, eh, el, direction = self.types[m.group(2)]
                for nm in m.group(1).split(","):
                    if direction == "to":   # a memory: array (0 to n-1)
                        self.mem[nm.strip()] = (eh - el + 1, {})
                    else:
                        self.arr[nm.strip()] = (hi, lo, [Wire(eh, el) for _ in range(hi - lo + 1)])
                continue
            assert not st.strip() or st.startswith(("type ", "variable ")), "unparsed declaration in %s: %r" % (self.ent.name, st[:100])

    def _gen_body(self, inner, env):
        if re.match(r"\s*(signal|constant|type|function) ", inner):
            inner = R._take_functions(inner, self.funcs)
            k = re.match(r"(.*?)\bbegin (.*)$", inner)
            self._decls(k.group(1).strip(), env)
            return k.group(2)
        return re.sub(r"^ ?begin ", "", inner)

    def _region(self, text, env):
        for st in R._statements(text):
            m = re.match(r"(\w+) ?: ?if (.*?) generate (.*) end generate(?: \w+)?$", st)
            if m:
                if R._cond(m.group(2), env):
                    self._region(self._gen_body(m.group(3), env), env)
                continue
            m = re.match(r"(\w+) ?: ?for (\w+) in (.*?) to (.*?) generate (.*) end generate(?: \w+)?$", st)
            if m:
                for v in range(R._int(m.group(3), env), R._int(m.group(4), env) + 1):
                    e2 = dict(env, **{m.group(2): v})
                    self._region(self._gen_body(m.group(5), e2), e2)
                continue
            m = re.match(r"(\w+) ?: ?(entity work\.\w+|dsp48e1|dsp48e2) ?(?:generic map ?\((.*?)\) ?)?port map ?\((.*)\)$", st)
            if m:
                gmap = {k.strip(): v.strip() for k, v in (x.split("=>", 1) for x in R._split_top(m.group(3), ","))} if m.group(3) else {}
                pmap = {k.strip(): v.strip() for k, v in (x.split("=>", 1) for x in R._split_top(m.group(4), ","))}
                unit = m.group(2).replace("entity work.", "")
                if unit in ("dsp48e1", "dsp48e2"):
                    self.kids.append(Dsp("E1" if unit.endswith("1") else "E2", gmap, pmap, env, self))
                else:
                    g = {}
                    for k, v in gmap.items():
                        g[k] = (v.strip('"') if v.startswith('"') else {"true": True, "false": False}[v] if v in ("true", "false") else
                                (env[v] if (v in env and not isinstance(env[v], int)) else R._int(v, env)))
                    self.kids.append(Node(unit, g, self, pmap, env))
                continue
            m = re.match(r"(\w+) ?: ?process ?\(.*?\) ?(?:is )?begin (.*) end process(?: \w+)?$", st)
            if m:
                body = m.group(2).strip()
                k = re.match(r"if (?:\(? ?rising_edge ?\( ?clk ?\) ?\)?|\( ?clk'event and clk ?= ?'1' ?\)) then (.*) end if ?;?$", body)
                assert k, "a process that is not clocked: %r" % body[:80]
                self.procs.append((R._parse_seq(k.group(1)), env))
                continue
            m = re.match(r"([\w]+(?: ?\(.*?\))*) ?<= ?(.*)$", st)
            assert m, "unparsed statement in %s: %r" % (self.ent.name, st[:120])
            rhs = re.sub(r"\s*\bafter [\w.]+( ns\b)?", "", m.group(2))
            reg = bool(re.search(r"\bwhen rising_edge ?\( ?clk ?\)", rhs))
            en = re.search(r"\bwhen rising_edge ?\( ?clk ?\) and (.*)$", rhs)
            rhs = re.sub(r"\s*\bwhen rising_edge ?\( ?clk ?\)( and .*)?$", "", rhs).strip()
            (self.regs if reg else self.comb).append((m.group(1).strip(), rhs, env, en.group(1).strip() if en else None))

    # ---- references and values -----------------------------------------------------------------------------------------------------
    def _attr(self, text):
        def left(m):
            nm = m.group(1)
            return str(self.arr[nm][0] if nm in self.arr else self.w[nm].hi)
        return re.sub(r"(\w+)'left", left, text)

    def ref(self, text, env):
        """-> ("w", Wire, hi, lo) | ("a", name, hi, lo) for a slice of an array of vectors (whole elements); indices are elaboration constants"""
        key = (text, id(env))
        r = self._rc.get(key)
        if r is None:
            self._keep[id(env)] = env
            r = self._rc[key] = self._ref(text, env)
        return r

    def _ref(self, text, env):
        text = self._attr(text.strip())
        m = re.match(r"(\w+) ?(?:\(([^()]*(?:\([^()]*\)[^()]*)*)\))? ?(?:\(([^()]*(?:\([^()]*\)[^()]*)*)\))?$", text)
        assert m, "unparsed reference %r" % text
        name, i1, i2 = m.group(1), m.group(2), m.group(3)
        if name in self.arr:
            hi, lo, els = self.arr[name]
            if i1 is None:
                return ("a", name, hi, lo)
            if " downto " in i1:
                a, b = i1.split(" downto ")
                return ("a", name, R._int(a, env), R._int(b, env))
            el = els[R._int(i1, env) - lo]
            if i2 is None:
                return ("w", el, el.hi, el.lo)
            a, b = i2.split(" downto ")
            return ("w", el, R._int(a, env), R._int(b, env))
        s = self.w[name]
        assert i2 is None
        if i1 is None:
            return ("w", s, s.hi, s.lo)
        if " downto " in i1:
            a, b = i1.split(" downto ")
            return ("w", s, R._int(a, env), R._int(b, env))
        i = R._int(i1, env)
        return ("w", s, i, i)

    def value(self, text, env, want_w=None):
        """-> (value, width) of an expression now.  The text is parsed once per (expression, region, wanted width) into a closure."""
        key = (text, id(env), want_w)
        f = self._vc.get(key)
        if f is None:
            self._keep[id(env)] = env   # the key holds the region's identity: the region must stay alive
            f = self._vc[key] = self._compile(text, env, want_w)
        return f()

    def _compile(self, text, env, want_w):
        text = self._attr(text.strip())
        comp = self._compile
        while text.startswith("(") and text.endswith(")") and R._balanced(text[1:-1]) and "=>" not in text:
            text = text[1:-1].strip()
        parts = R._split_top(text, "&")
        if len(parts) > 1:
            fs = [comp(part, env, None) for part in parts]

            def cat():
                v = w = 0
                for f in fs:
                    pv, pw = f()
                    v, w = (v << pw) | pv, w + pw
                return v, w
            return cat
        parts = R._split_kw(text, " and ")
        if len(parts) > 1:
            f0, rest = comp(parts[0], env, want_w), [comp(x, env, want_w) for x in parts[1:]]

            def conj():
                v, w = f0()
                for g in rest:
                    v &= g()[0]
                return v, w
            return conj
        parts = R._split_kw(text, " or ")
        if len(parts) > 1:
            f0, rest = comp(parts[0], env, want_w), [comp(x, env, want_w) for x in parts[1:]]

            def disj():
                v, w = f0()
                for g in rest:
                    v |= g()[0]
                return v, w
            return disj
        parts = R._split_top(text, "*")
        if len(parts) == 2 and all(x.startswith("unsigned") for x in parts):
            fa, fb = (comp(re.match(r"unsigned ?\((.*)\)$", x).group(1), env, None) for x in parts)

            def mul():
                (a, wa), (b, wb) = fa(), fb()
                return a * b, wa + wb
            return mul
        m = re.match(r"(\w+) ?\( ?conv_integer ?\( ?(?:unsigned ?\()?(\w+)\)? ?\) ?\)$", text)
        if m and isinstance(env.get(m.group(1)), dict):
            rom, fi = env[m.group(1)], comp(m.group(2), env, None)
            return lambda: (rom[fi()[0]], want_w)
        if m and m.group(1) in self.mem:
            w, mem = self.mem[m.group(1)]
            fi = comp(m.group(2), env, None)
            return lambda: (mem.get(fi()[0], 0), w)
        if text.startswith('x"'):
            c = (int(text[2:-1], 16), 4 * (len(text) - 3))
            return lambda: c
        m = re.match(r"(\w+) ?\((.*)\)$", text)
        if m and m.group(1) in self.funcs:   # a function of the entity over a vector (int_bitrev_order's bit_pair): executed, not restated
            fname, argt = m.group(1), [a.strip() for a in R._split_top(m.group(2), ",")]

            def call():
                args = []
                for a in argt:
                    if a in self.w:
                        args.append((lambda v: (lambda i: (v >> int(i)) & 1))(self.w[a].val))
                    else:
                        args.append(env[a] if a in env else R._int(a, env))
                bits = R.call_function(self.funcs, fname, args, env)
                return sum(int(b) << i for i, b in bits.items()), (max(bits) + 1 if want_w is None else want_w)
            return call
        parts = R._split_top(text, "+")
        if len(parts) >= 2:   # cnt + '1', cnt_even + wr_inz + 1, in_cnt + 1
            f0 = comp(parts[0], env, want_w)
            lits = sum(int(x.strip().strip("'")) for x in parts[1:] if re.match(r"'?\d+'?$", x.strip()))
            others = [x.strip() for x in parts[1:] if not re.match(r"'?\d+'?$", x.strip())]

            def add():
                v, w = f0()
                v += lits
                for x in others:
                    v += self.value(x, env, w)[0]
                return v & ((1 << w) - 1), w
            return add
        m = re.match(r"(\w+) ?\( ?(\w+) ?\)$", text)
        if m and isinstance(env.get(m.group(1)), dict):   # a constant array: std_inc(0), inc_bit(in_cnt)
            table = env[m.group(1)]
            if m.group(2) in self.ints:
                idx = self.w[m.group(2)]
                return lambda: (table[idx.val], want_w)
            c = (table[R._int(m.group(2), env)], want_w)
            return lambda: c
        if m and m.group(1) in self.w and m.group(2) in self.ints:   # sw_ptr(wr_rng)
            vec, idx = self.w[m.group(1)], self.w[m.group(2)]
            return lambda: ((vec.val >> idx.val) & 1, 1)
        if re.match(r"\d+$", text):
            c = (int(text), want_w)
            return lambda: c
        m = re.match(r"not ?\(([^()]*(?:\([^()]*\)[^()]*)*)\)$", text) or re.match(r"not (.+)$", text)
        if m:
            f = comp(m.group(1), env, want_w)

            def inv():
                v, w = f()
                return (~v) & ((1 << w) - 1), w
            return inv
        m = re.match(r"\( ?others ?=> ?(.*)\)$", text)
        if m:
            inner = m.group(1).strip()
            if inner.startswith("("):  # (others => (others => '0')): a whole array
                return lambda: (0, want_w)
            if inner.startswith("'"):
                c = (((1 << want_w) - 1 if int(inner.strip("'")) else 0), want_w)
                return lambda: c
            f = comp(inner, env, None)
            ones = (1 << want_w) - 1
            return lambda: ((ones if f()[0] else 0), want_w)
        m = re.match(r"\((.*others.*)\)$", text)
        if m:
            v = 0
            for it in R._split_top(m.group(1), ","):
                k, b = [x.strip() for x in it.split("=>")]
                if k != "others" and b == "'1'":
                    v |= 1 << int(k)
            c = (v, want_w)
            return lambda: c
        m = re.match(r"sxt ?\((.*), ?([^,]+)\)$", text)
        if m:
            f, n = comp(m.group(1), env, None), R._int(m.group(2), env)

            def ext():
                v, w = f()
                return tw.sxt(v, w, n), n
            return ext
        if text.startswith('"'):
            c = (int(text.strip('"'), 2), len(text) - 2)
            return lambda: c
        if text.startswith("'"):
            c = (int(text.strip("'")), 1)
            return lambda: c
        r = self.ref(text, env)
        assert r[0] == "w", "an array where a vector is expected: %r" % text
        wire, hi, lo = r[1], r[2], r[3]
        if hi < lo:
            return lambda: (0, hi - lo + 1)
        assert wire.lo <= lo <= hi <= wire.hi, "slice (%d downto %d) outside (%d downto %d)" % (hi, lo, wire.hi, wire.lo)
        sh, mask, w = lo - wire.lo, (1 << (hi - lo + 1)) - 1, hi - lo + 1
        return lambda: ((wire.val >> sh) & mask, w)

    def array_value(self, text, env):
        """RHS of an assignment to an array of vectors: slice & element & ... -> list of element values, highest index first"""
        out = []
        for part in R._split_top(self._attr(text), "&"):
            r = self.ref(part, env) if re.match(r"\w+", part.strip()) and part.strip().split("(")[0].strip() in self.arr else None
            if r and r[0] == "a":
                hi0, lo0, els = self.arr[r[1]]
                out += [els[i - lo0].val for i in range(r[2], r[3] - 1, -1)]
            else:
                out.append(self.value(part, env)[0])
        return out

    def assign(self, lhs, env, v):
        r = self.ref(lhs, env)
        assert r[0] == "w"
        return r[1].set(r[2], r[3], v)

    def cond(self, c, env):
        key = (c, id(env))
        f = self._cc.get(key)
        if f is None:
            self._keep[id(env)] = env
            f = self._cc[key] = self._compile_cond(c, env)
        return f()

    def _compile_cond(self, c, env):
        c = c.strip()
        while c.startswith("(") and c.endswith(")") and R._balanced(c[1:-1]):
            c = c[1:-1].strip()
        for op_, fn in ((" or ", any), (" and ", all)):
            parts = R._split_kw(c, op_)
            if len(parts) > 1:
                fs = [self._compile_cond(x, env) for x in parts]
                return (lambda fs=fs: any(f() for f in fs)) if fn is any else (lambda fs=fs: all(f() for f in fs))
        m = re.match(r"(.*?) ?= ?'([01])'$", c)
        if m:
            f, bit = self._compile(m.group(1), env, None), int(m.group(2))
            return lambda: f()[0] == bit
        m = re.match(r"(\w+) ?= ?(.+)$", c)
        assert m and m.group(1) in self.ints, "unparsed condition %r" % c
        wire, k = self.w[m.group(1)], R._int(m.group(2), env)
        return lambda: wire.val == k

    # ---- simulation ----------------------------------------------------------------------------------------------------------------
    def _closure(self, text, env, want_w):
        key = (text, id(env), want_w)
        f = self._vc.get(key)
        if f is None:
            self._keep[id(env)] = env
            f = self._vc[key] = self._compile(text, env, want_w)
        return f

    @staticmethod
    def _slot(wire, hi, lo):
        """(wire, shift, mask, ~(mask << shift)) of wire(hi downto lo)"""
        assert wire.lo <= lo <= hi <= wire.hi, "slice (%d downto %d) outside (%d downto %d)" % (hi, lo, wire.hi, wire.lo)
        sh, mask = lo - wire.lo, (1 << (hi - lo + 1)) - 1
        return wire, sh, mask, ~(mask << sh)

    def _prepare(self):
        """the combinational statements, port connections and DSP outputs of this node as (target slot, closure) lists: looked up once"""
        self._cw, self._ca, self._kc = [], [], []
        for lhs, rhs, env, _ in self.comb:
            r = self.ref(lhs, env)
            if r[0] == "a":
                self._ca.append((r, rhs, env))
            elif r[2] >= r[3]:
                self._cw.append(self._slot(r[1], r[2], r[3]) + (self._closure(rhs, env, r[2] - r[3] + 1),))
        for k in self.kids:
            if isinstance(k, Dsp):
                outs = []
                for port, attr in (("p", "p"), ("pcout", "p"), ("carrycascout", "cy")):
                    a = k.pmap.get(port)
                    if a and a != "open":
                        r = self.ref(a, k.env)
                        assert r[0] == "w"
                        if r[2] >= r[3]:
                            outs.append(self._slot(r[1], r[2], r[3]) + (attr,))
                self._kc.append((k, None, outs))
                continue
            ins, outs = [], []
            for port, (d, _) in k.ent.ports.items():
                if d == "in" and port in k.pmap:
                    pw = k.w[port]
                    ins.append((pw, (1 << pw.width) - 1, self._closure(k.pmap[port], k.penv, pw.width)))
                elif d == "out" and port in k.pmap and k.pmap[port] != "open":
                    r = self.ref(k.pmap[port], k.penv)
                    assert r[0] == "w"
                    if r[2] >= r[3]:
                        outs.append(self._slot(r[1], r[2], r[3]) + (k.w[port],))
            self._kc.append((k, ins, outs))

    def settle_once(self):
        if self._cw is None:
            self._prepare()
        ch = False
        for wire, sh, mask, nm, f in self._cw:
            old = wire.val
            new = (old & nm) | ((f()[0] & mask) << sh)
            if new != old:
                wire.val = new
                ch = True
        for r, rhs, env in self._ca:
            hi0, lo0, els = self.arr[r[1]]
            for i, v in zip(range(r[2], r[3] - 1, -1), self.array_value(rhs, env)):
                ch |= els[i - lo0].set(els[i - lo0].hi, els[i - lo0].lo, v)
        for k, ins, outs in self._kc:
            if ins is None:   # a DSP48: its registered outputs onto this node's signals
                for wire, sh, mask, nm, attr in outs:
                    old = wire.val
                    new = (old & nm) | ((getattr(k, attr) & mask) << sh)
                    if new != old:
                        wire.val = new
                        ch = True
                continue
            for pw, mask, f in ins:      # parent -> child inputs
                v = f()[0] & mask
                if v != pw.val:
                    pw.val = v
                    ch = True
            if k.settle_once():
                ch = True
            for wire, sh, mask, nm, src in outs:   # child outputs -> parent
                old = wire.val
                new = (old & nm) | ((src.val & mask) << sh)
                if new != old:
                    wire.val = new
                    ch = True
        return ch

    def settle(self):
        for _ in range(64):
            if not self.settle_once():
                return
        raise AssertionError("combinational logic does not settle")

    def _seq(self, seq, env, acts, writes):
        for node in seq:
            if node[0] == "assign":
                m = re.match(r"(\w+) ?\( ?conv_integer ?\( ?(\w+) ?\) ?\)$", node[1])
                if m and m.group(1) in self.mem:
                    writes.append((self.mem[m.group(1)][1], self.value(m.group(2), env)[0], self.value(node[2], env, self.mem[m.group(1)][0])[0]))
                    continue
                r = self.ref(node[1], env)
                if r[0] == "a":
                    hi0, lo0, els = self.arr[r[1]]
                    for i, v in zip(range(r[2], r[3] - 1, -1), self.array_value(node[2], env)):
                        acts.append((els[i - lo0], els[i - lo0].hi, els[i - lo0].lo, v))
                else:
                    acts.append((r[1], r[2], r[3], self.value(node[2], env, r[2] - r[3] + 1)[0]))
            else:
                for c, body in node[1]:
                    if self.cond(c, env):
                        self._seq(body, env, acts, writes)
                        break
                else:
                    self._seq(node[2], env, acts, writes)

    def compute(self, acts, writes):
        for lhs, rhs, env, en in self.regs:
            if en is not None and not self.cond(en, env):
                continue
            r = self.ref(lhs, env)
            if r[0] == "a":
                hi0, lo0, els = self.arr[r[1]]
                for i, v in zip(range(r[2], r[3] - 1, -1), self.array_value(rhs, env)):
                    acts.append((els[i - lo0], els[i - lo0].hi, els[i - lo0].lo, v))
            else:
                acts.append((r[1], r[2], r[3], self.value(rhs, env, r[2] - r[3] + 1)[0]))
        for ast, env in self.procs:
            self._seq(ast, env, acts, writes)
        for k in self.kids:
            if isinstance(k, Dsp):
                k.compute()
            else:
                k.compute(acts, writes)

    def commit_dsps(self):
        for k in self.kids:
            if isinstance(k, Dsp):
                k.commit()
            else:
                k.commit_dsps()

    def clock(self, inputs):
        for k, v in inputs.items():
            s = self.w[k]
            s.set(s.hi, s.lo, v)
        self.settle()
        acts, writes = [], []
        self.compute(acts, writes)
        for s, hi, lo, v in acts:
            s.set(hi, lo, v)
        for mem, idx, v in writes:
            mem[idx] = v
        self.commit_dsps()
        self.settle()
        return {p: self.w[p].val for p, (d, _) in self.ent.ports.items() if d == "out"}


def run_core(direction, nfft, dw, tw_, fmt, rnd, xser, frames, ramb="wrap", use_fly=1, gap=0, use_mlt=False):
    """frames: list of frames, each a list of N (re, im) -- natural order for FWD (lane 0 = x[i], lane 1 = x[i + N/2]), the bit-reversed pair
    stream for INV (lane 0 = v[2i], lane 1 = v[2i + 1]).  -> list of output beats ((re0, im0), (re1, im1)) that came with DO_VAL = '1'"""
    top = Node("int_fftnk" if direction == "FWD" else "int_ifftnk",
               {"nfft": nfft, "ramb_type": ramb, "format": fmt, "rndmode": rnd, "data_width": dw, "twdl_width": tw_, "xser": xser.lower(), "use_mlt": use_mlt})
    n = 1 << nfft
    ow = dw + fmt * nfft
    idle = {"rst": 0, "use_fly": use_fly, "di_ena": 0, "di_re0": 0, "di_im0": 0, "di_re1": 0, "di_im1": 0}
    for _ in range(4):
        top.clock(dict(idle, rst=1))
    for _ in range(4):
        top.clock(idle)
    beats = []

    def tick(inp):
        o = top.clock(inp)
        if o["do_val"]:
            beats.append(((tw.signed(o["do_re0"], ow), tw.signed(o["do_im0"], ow)), (tw.signed(o["do_re1"], ow), tw.signed(o["do_im1"], ow))))

    for fr in frames:
        for i in range(n // 2):
            a, b = (fr[i], fr[i + n // 2]) if direction == "FWD" else (fr[2 * i], fr[2 * i + 1])
            tick(dict(idle, di_ena=1, di_re0=tw.vec(a[0], dw), di_im0=tw.vec(a[1], dw), di_re1=tw.vec(b[0], dw), di_im1=tw.vec(b[1], dw)))
        for _ in range(gap):
            tick(idle)
    for _ in range(40 * nfft + 4 * n):
        tick(idle)
        if ramb == "cont" and len(beats) >= len(frames) * n // 2 + 8:   # every frame is out (and a few clocks more show that nothing follows)
            break
    return beats, top


def _reset(top, idle, rst):
    for _ in range(4):
        top.clock(dict(idle, **{rst: 1}))
    for _ in range(4):
        top.clock(idle)


def run_single_path(nfft, dw, tw_, fmt, rnd, xser, frames, fly=1, gap=0, flush=0, lead=4):
    """int_fft_single_path (main/int_fft_single_path.vhd) from the text: one sample per clock in natural order through inbuf_half_path ->
    int_fftNk (CONT) -> outbuf_half_path -> two int_bitrev_order.  -> the DO_VL-qualified samples [(re, im)], natural order, frame after frame
    (the bit-reverse buffer hands a frame out while the next one comes in: the last frame stays inside unless `flush` all-zero frames follow).
    gap: idle clocks after every frame (the protocol of the kit's testbenches, tools/vivado_crosscheck/tb_single_*.vhd)."""
    top = Node("int_fft_single_path", {"nfft": nfft, "data_width": dw, "twdl_width": tw_, "format": fmt, "rndmode": rnd, "xseries": xser.lower(),
                                       "use_mlt": False})
    ow = dw + fmt * nfft
    n = 1 << nfft
    idle = {"reset": 0, "fly_fwd": fly, "di_en": 0, "di_re": 0, "di_im": 0}
    _reset(top, idle, "reset")
    for _ in range(lead):
        top.clock(idle)
    out = []
    quiet = [0]

    def tick(inp):
        o = top.clock(inp)
        quiet[0] += 1
        if o["do_vl"]:
            quiet[0] = 0
            out.append((tw.signed(o["do_re"], ow), tw.signed(o["do_im"], ow)))

    for fr in list(frames) + [[(0, 0)] * n] * flush:
        for a, b in fr:
            tick(dict(idle, di_en=1, di_re=tw.vec(int(a), dw), di_im=tw.vec(int(b), dw)))
        for _ in range(gap):
            tick(idle)
    for _ in range(8 * n + 4096):   # the testbenches' drain; nothing can follow 2N + 40 NFFT + 200 silent idle clocks
        tick(idle)
        if quiet[0] > 2 * n + 40 * nfft + 200:
            break
    return out, top


def run_pair(nfft, dw, tw_, fmt, rnd, xser, frames, ramb="cont", gap=0, flush=0, toggle=False, ports=False, lead=4):
    """int_fft_ifft_pair (main/int_fft_ifft_pair.vhd) from the text: two samples per clock, lane 0 = x[2i], lane 1 = x[2i + 1], through
    iobuf_flow_int2 (CONT) or iobuf_wrap_int2 (WRAP) -> int_fftNk -> int_ifftNk -> the same buffer in its BITREV form.
    -> beats ((re0, im0), (re1, im1)) read from the wrapper's own dt_rev0 / dt_rev1 (its output ports duplicate slices: Q0_IM = Q0_RE,
    Q1_RE = Q1_IM, SURVEY 9.9 -- asserted here), or with ports=True the four port values (Q0_RE, Q1_RE, Q0_IM, Q1_IM) as a testbench sees them.
    toggle: the enable of every beat followed by one idle clock (fft_double_test.vhd with RAMB_TYPE = "WRAP"; the kit's tb_pair_*.vhd)."""
    top = Node("int_fft_ifft_pair", {"nfft": nfft, "ramb_type": ramb, "data_width": dw, "twdl_width": tw_, "format": fmt, "rndmode": rnd,
                                     "xseries": xser.lower(), "use_mlt": False})
    ow = dw + fmt * 2 * nfft
    idle = {"reset": 0, "fly_fwd": 1, "fly_inv": 1, "di_en": 0, "d0_re": 0, "d0_im": 0, "d1_re": 0, "d1_im": 0}
    _reset(top, idle, "reset")
    for _ in range(lead):
        top.clock(idle)
    out = []
    mask = (1 << ow) - 1
    quiet = [0]

    def tick(inp):
        o = top.clock(inp)
        quiet[0] += 1
        if o["qo_vl"]:
            quiet[0] = 0
            r0, r1 = top.w["dt_rev0"].val, top.w["dt_rev1"].val
            assert o["q0_re"] == o["q0_im"] == r0 & mask and o["q1_re"] == o["q1_im"] == r1 >> ow, "the wrapper's ports are not the duplicated slices"
            if ports:
                out.append(tuple(tw.signed(o[k], ow) for k in ("q0_re", "q1_re", "q0_im", "q1_im")))
            else:
                out.append(((tw.signed(r0 & mask, ow), tw.signed(r0 >> ow, ow)), (tw.signed(r1 & mask, ow), tw.signed(r1 >> ow, ow))))

    n = 1 << nfft
    for fr in list(frames) + [[(0, 0)] * n] * flush:
        for i in range(n // 2):
            a, b = fr[2 * i], fr[2 * i + 1]
            beat = dict(idle, d0_re=tw.vec(int(a[0]), dw), d0_im=tw.vec(int(a[1]), dw), d1_re=tw.vec(int(b[0]), dw), d1_im=tw.vec(int(b[1]), dw))
            tick(dict(beat, di_en=1))
            if toggle:
                tick(beat)   # the data stay, the enable drops for one clock
        for _ in range(gap):
            tick(idle)
    for _ in range(8 * n + 8192):
        tick(idle)
        if quiet[0] > 2 * n + 80 * nfft + 200:
            break
    return out, top


def expected_natural(direction, nfft, dw, tw_, fmt, rnd, xser, frames):
    """oracle_py, natural order in and out, frame after frame: [(re, im)]"""
    from oracle import oracle_py as op
    ow = dw + fmt * nfft * (2 if direction == "PAIR" else 1)
    out = []
    for fr in frames:
        y = op.execute(fr, nfft, dw, tw_, fmt, rnd, xser == "NEW", {"FWD": op.FWD, "INV": op.INV, "PAIR": op.PAIR}[direction])
        out += [(op.sgn(a, ow), op.sgn(b, ow)) for a, b in y]
    return out


def compare_wrapper(which, nfft, dw, tw_, fmt, rnd, xser, count=4, seed=5, flush=True):
    """which: "single" | "pair" (RAMB_TYPE CONT) | "pair_wrap" (RAMB_TYPE WRAP driven like fft_double_test.vhd: toggling enable, 32 idle clocks
    between frames).  -> (equal?, whole frames out, frames in).  Without `flush` the buffers keep the last frames (they hand a frame out while
    later ones come in): at least one whole frame must come out and every sample that does must be the oracle's; with it (all-zero frames
    behind the data: 1 / 2 / 4) every frame must come out."""
    import random
    frames = _frames(random.Random(seed), nfft, dw, count)
    if which == "single":
        got, _ = run_single_path(nfft, dw, tw_, fmt, rnd, xser, frames, flush=1 if flush else 0)
        want = expected_natural("FWD", nfft, dw, tw_, fmt, rnd, xser, frames)
    else:
        wrap = which == "pair_wrap"
        beats, _ = run_pair(nfft, dw, tw_, fmt, rnd, xser, frames, "wrap" if wrap else "cont", gap=32 if wrap else 0,
                            flush=(4 if wrap else 2) if flush else 0, toggle=wrap)
        got = [s for b in beats for s in b]
        want = expected_natural("PAIR", nfft, dw, tw_, fmt, rnd, xser, frames)
    whole = min(count, len(got) >> nfft)
    k = min(len(got), len(want))
    return (whole >= (count if flush else 1) and got[:k] == want[:k]), whole, count


def expected(direction, nfft, dw, tw_, fmt, rnd, xser, frames, use_fly=1):
    from oracle import oracle_py as op
    out = []
    for fr in frames:
        if direction == "FWD":
            v = op.fft_dif(fr, nfft, dw, tw_, fmt, rnd, xser == "NEW", use_fly)
            out += [(v[2 * i], v[2 * i + 1]) for i in range(len(v) // 2)]
        else:
            y = op.ifft_dit(fr, nfft, dw, tw_, fmt, rnd, xser == "NEW", use_fly)
            h = len(y) // 2
            out += [(y[i], y[i + h]) for i in range(h)]
    ow = dw + fmt * nfft
    return [((op.sgn(a[0], ow), op.sgn(a[1], ow)), (op.sgn(b[0], ow), op.sgn(b[1], ow))) for a, b in out]


def _frames(rng, nfft, dw, count):
    lo, hi = -(1 << (dw - 1)), (1 << (dw - 1)) - 1
    fr = [[(rng.randint(lo, hi), rng.randint(lo, hi)) for _ in range(1 << nfft)] for _ in range(count)]
    if count > 1:   # one frame of the corners: full scale of both signs
        fr[-1] = [((lo, hi), (hi, lo), (lo, lo), (hi, hi))[i & 3] for i in range(1 << nfft)]
    return fr


def compare(direction, nfft, dw, tw_, fmt, rnd, xser, ramb="wrap", use_fly=1, gap=0, count=3, seed=7):
    """-> (equal?, beats out, beats expected).  RAMB_TYPE = "CONT" must give every beat; "WRAP" (the delay lines drain only while frames
    come in: int_delay_line / int_delay_wrap) may hold back the tail of the last frame -- what did come out must be the oracle's prefix."""
    import random
    frames = _frames(random.Random(seed), nfft, dw, count)
    got, _ = run_core(direction, nfft, dw, tw_, fmt, rnd, xser, frames, ramb, use_fly, gap)
    want = expected(direction, nfft, dw, tw_, fmt, rnd, xser, frames, use_fly)
    if ramb == "cont":
        ok = got == want
    else:
        ok = got[:len(want)] == want[:len(got)] and len(got) >= len(want) - (1 << (nfft - 1))
    return ok, len(got), len(want)


# The cases of profiles/r*_rtl_sim.txt (direction, NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSER, RAMB_TYPE, USE_FLY, gap)
SWEEP = (
    [(d, n, 16, 16, f, r, x, "wrap", 1, 0) for n in (3, 4) for d in ("FWD", "INV") for (f, r) in ((0, 0), (0, 1), (1, 0)) for x in ("NEW", "OLD")]
    + [("FWD", 3, 16, 16, 0, 0, "NEW", "cont", 1, 0), ("INV", 3, 16, 16, 1, 0, "OLD", "cont", 1, 0), ("INV", 4, 16, 16, 0, 1, "NEW", "cont", 1, 0),
       ("FWD", 3, 16, 16, 0, 0, "NEW", "cont", 1, 5), ("INV", 3, 16, 16, 0, 1, "OLD", "cont", 1, 3),     # idle clocks between frames
       ("FWD", 3, 16, 16, 0, 0, "NEW", "cont", 0, 0), ("INV", 3, 16, 16, 1, 0, "NEW", "cont", 0, 0),     # USE_FLY = 0: the bypass mux
       ("FWD", 3, 12, 10, 0, 0, "NEW", "wrap", 1, 0),
       ("FWD", 3, 24, 24, 1, 0, "NEW", "wrap", 1, 0), ("INV", 3, 24, 24, 1, 0, "NEW", "wrap", 1, 0), ("FWD", 3, 24, 24, 0, 1, "OLD", "cont", 1, 0),
       ("FWD", 3, 30, 16, 1, 0, "NEW", "wrap", 1, 0), ("FWD", 3, 30, 16, 1, 0, "OLD", "wrap", 1, 0), ("INV", 3, 32, 16, 0, 0, "NEW", "cont", 1, 0),
       ("FWD", 3, 40, 24, 1, 0, "NEW", "cont", 1, 0), ("FWD", 3, 46, 16, 1, 0, "NEW", "wrap", 1, 0), ("INV", 3, 52, 16, 1, 0, "OLD", "cont", 1, 0),
       ("FWD", 5, 16, 16, 0, 0, "NEW", "cont", 1, 0), ("INV", 5, 16, 16, 0, 1, "OLD", "cont", 1, 0)])

# ADD_DELAY of the butterflies = addsub_delay(DTW + SCALE + RNDMODE) + RNDMODE (int_dif2_fly.vhd / int_dit2_fly.vhd), while the adder they
# instantiate is built for DSPW = DTW - 1 (scaled truncate) / DTW (scaled round, unscaled): the two disagree on "below 48 bits" at scaled
# DTW = 46 (round), 47, 48 (truncate), the valid strobe leaves one clock away from the data and the frames of the TEXT are wrong there.  Outside the documented DATA_WIDTH range
# (8 .. 32); the oracle and the engine compute the intended arithmetic.  (direction, DTW, RNDMODE, FORMAT, XSER) -> does the text agree?
STROBE_CORNER = (("FWD", 45, 1, 0, "NEW", True), ("FWD", 46, 0, 0, "NEW", True), ("FWD", 46, 1, 0, "NEW", False), ("FWD", 47, 0, 0, "NEW", False),
                 ("INV", 47, 1, 0, "OLD", False), ("FWD", 48, 0, 0, "NEW", False), ("INV", 48, 0, 0, "OLD", False), ("FWD", 48, 1, 0, "NEW", True),
                 ("FWD", 49, 0, 0, "NEW", True), ("FWD", 47, 0, 1, "NEW", True), ("FWD", 48, 0, 1, "NEW", True))


WRAPPERS = ([("single", n, 16, 16, f, r, x) for n in (3, 4) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "OLD"), (1, 0, "NEW"))]
            + [("single", 5, 16, 16, 0, 0, "NEW"), ("single", 3, 24, 24, 1, 0, "OLD"), ("single", 7, 16, 16, 0, 1, "NEW"), ("single", 10, 16, 16, 0, 0, "NEW")]
            + [("pair", n, 16, 16, f, r, x) for n in (3, 4) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "NEW"), (1, 0, "OLD"))]
            + [("pair", 5, 16, 16, 0, 0, "NEW"), ("pair", 7, 16, 16, 0, 0, "NEW"), ("pair_wrap", 3, 16, 16, 0, 0, "NEW"), ("pair_wrap", 5, 16, 16, 0, 1, "NEW"),
               ("pair_wrap", 7, 16, 16, 1, 0, "OLD")])
# longer frames: N = 1024 (BASELINE's C2 shape), N = 4096 / 8192 where STAGE 11 / 12 take their twiddles from row_twiddle_tay, the 24-bit
# unscaled regime walk of C3 at N = 4096
LONG = [("FWD", 6, 16, 16, 0, 1, "NEW", "cont", 1, 0), ("INV", 7, 16, 16, 0, 0, "OLD", "wrap", 1, 0), ("FWD", 10, 16, 16, 0, 0, "NEW", "cont", 1, 0),
        ("INV", 10, 16, 16, 0, 1, "OLD", "cont", 1, 0), ("FWD", 12, 16, 16, 0, 0, "NEW", "cont", 1, 0), ("INV", 12, 16, 16, 0, 0, "NEW", "cont", 1, 0),
        ("FWD", 12, 16, 16, 0, 1, "OLD", "cont", 1, 0), ("FWD", 12, 24, 24, 1, 0, "NEW", "cont", 1, 0), ("FWD", 13, 16, 16, 0, 0, "NEW", "cont", 1, 0)]


def sweep():
    import time
    print("# tools/rtl_sim.py --sweep: int_fftNk / int_ifftNk elaborated from the reference's own VHDL text, clocked beat by beat, against oracle_py")
    bad = 0
    for (d, n, dw, t, f, r, x, ramb, fly, gap) in SWEEP + LONG:
        t0 = time.time()
        ok, a, b = compare(d, n, dw, t, f, r, x, ramb, fly, gap, count=3 if n < 10 else 2 if n < 12 else 1)
        print("%s NFFT %2d DW %2d TW %2d FORMAT %d RNDMODE %d %s RAMB %s USE_FLY %d gap %d: %3d of %3d beats, %s  (%.0f s)"
              % (d, n, dw, t, f, r, x, ramb.upper(), fly, gap, a, b, "equal" if ok else "DIFFERENT", time.time() - t0), flush=True)
        bad += not ok
    print("# USE_MLT = TRUE (rom_twiddle_int computes MATHPI * cnt with a multiplier instead of the accumulating form): N = 4096, STAGE 11")
    import random
    for (d, x) in (("FWD", "NEW"), ("INV", "OLD")):
        t0 = time.time()
        fr = _frames(random.Random(77), 12, 16, 1)
        got, _ = run_core(d, 12, 16, 16, 0, 0, x, fr, "cont", use_mlt=True)
        ok = got[:len(fr) << 11] == expected(d, 12, 16, 16, 0, 0, x, fr) and len(got) >= 2048
        print("%s NFFT 12 DW 16 TW 16 FORMAT 0 RNDMODE 0 %s USE_MLT TRUE: %d beats, %s  (%.0f s)" % (d, x, len(got), "equal" if ok else "DIFFERENT", time.time() - t0), flush=True)
        bad += not ok
    print("# the strobe corner: ADD_DELAY = addsub_delay(DTW+SCALE+RNDMODE)+RNDMODE against an adder of DSPW = DTW-1 (truncate) / DTW bits")
    for (d, dw, r, f, x, agree) in STROBE_CORNER:
        ok, a, b = compare(d, 3, dw, 16, f, r, x, "cont", count=2)
        print("%s NFFT  3 DW %2d TW 16 FORMAT %d RNDMODE %d %s: %s (predicted from the text: %s)"
              % (d, dw, f, r, x, "equal" if ok else "DIFFERENT", "equal" if agree else "DIFFERENT"), flush=True)
        bad += ok != agree
    print("# the wrappers of src/vhdl/main from the text, I/O buffers included, natural order in and out (memory order of the C-ABI's NATURAL);")
    print("# all-zero frames behind the data push the last frames out (1 single path, 2 pair CONT, 4 pair WRAP); WRAP as fft_double_test.vhd drives it")
    for (which, n, dw, t, f, r, x) in WRAPPERS:
        t0 = time.time()
        ok, whole, count = compare_wrapper(which, n, dw, t, f, r, x)
        print("%s NFFT %2d DW %2d TW %2d FORMAT %d RNDMODE %d %s: %d of %d frames out, %s  (%.0f s)"
              % ({"single": "int_fft_single_path     ", "pair": "int_fft_ifft_pair CONT  ", "pair_wrap": "int_fft_ifft_pair WRAP  "}[which], n, dw, t, f, r, x, whole, count,
                 "equal" if ok else "DIFFERENT", time.time() - t0), flush=True)
        bad += not ok
    print("rtl_sim: %d configurations, %d unexpected" % (len(SWEEP) + len(LONG) + 2 + len(STROBE_CORNER) + len(WRAPPERS), bad))
    return 1 if bad else 0


def fuzz(count, seed):
    """Random generics inside (and a little beyond) the documented ranges -- NFFT 3 .. 7, DATA_WIDTH 8 .. 44, TWDL_WIDTH 8 .. 26, the three
    modes, XSER, direction, RAMB_TYPE, now and then idle clocks between frames or USE_FLY = 0 -- each elaborated whole from the text and
    clocked against oracle_py.  Generics the oracle's validator refuses are skipped (counted); generics it accepts must elaborate."""
    import random
    import time
    from oracle import oracle_c as C
    rng = random.Random(seed)
    t0 = time.time()
    done = bad = refused = 0
    print("# tools/rtl_sim.py --fuzz %d %d" % (count, seed))
    while done < count:
        nfft = rng.choice([3, 3, 4, 4, 5, 5, 6, 7])
        dw = rng.choice([16, 16, rng.randint(8, 32), rng.randint(8, 32), rng.randint(33, 44)])
        t = rng.choice([16, 16, rng.randint(8, 26), rng.randint(8, 26)])
        fmt = rng.randint(0, 1)
        rnd = 0 if fmt else rng.randint(0, 1)
        xser = rng.choice(["NEW", "OLD"])
        d = rng.choice(["FWD", "INV"])
        ramb = rng.choice(["cont", "cont", "wrap"])
        gap = rng.choice([0, 0, 0, rng.randint(1, 9)]) if ramb == "cont" else 0
        fly = 0 if rng.random() < 0.08 else 1
        p = C.make_params(nfft, dw, t, fmt, rnd, xser == "NEW", fly)
        valid = C.lib().orc_validate(p, C.FWD if d == "FWD" else C.INV) == 0
        tag = "%s NFFT %d DW %2d TW %2d FORMAT %d RNDMODE %d %s RAMB %s USE_FLY %d gap %d" % (d, nfft, dw, t, fmt, rnd, xser, ramb.upper(), fly, gap)
        if not valid:   # no generate branch of int_cmult_dsp48 matches such widths: nothing drives the product (an unconnected net is not an
            refused += 1   # error of this simulation; that the accept sets agree is tests/test_rtl_interp.py's multiplier-tree check)
            continue
        try:
            ok, a, b = compare(d, nfft, dw, t, fmt, rnd, xser, ramb, fly, gap, count=2, seed=rng.randint(1, 1 << 30))
        except AssertionError as exc:
            print(tag, ": the oracle accepts it, the text does not elaborate:", str(exc)[:100])
            bad += 1
            R.forget()
            continue
        if not ok:
            print(tag, ": %d of %d beats, DIFFERENT" % (a, b), flush=True)
            bad += 1
        done += 1
    print("rtl_sim --fuzz: %d configurations clocked against the oracle, %d more refused by the oracle's validator, %d unexpected, %.0f s" % (done, refused, bad, time.time() - t0))
    return 1 if bad else 0


def fuzz_wrappers(count, seed):
    """The same for the wrappers of src/vhdl/main with their I/O buffers: int_fft_single_path, int_fft_ifft_pair with RAMB_TYPE CONT and WRAP
    (the latter driven like fft_double_test.vhd), random NFFT 3 .. 6, widths, mode, series; every frame must come out and be the oracle's."""
    import random
    import time
    from oracle import oracle_c as C
    rng = random.Random(seed)
    t0 = time.time()
    done = bad = refused = 0
    print("# tools/rtl_sim.py --fuzz-wrappers %d %d" % (count, seed))
    while done < count:
        which = rng.choice(["single", "pair", "pair_wrap"])
        nfft = rng.choice([3, 4, 4, 5, 5, 6])
        dw = rng.choice([16, 16, rng.randint(8, 32), rng.randint(8, 32)])
        t = rng.choice([16, 16, rng.randint(8, 26)])
        fmt = rng.randint(0, 1)
        rnd = 0 if fmt else rng.randint(0, 1)
        xser = rng.choice(["NEW", "OLD"])
        p = C.make_params(nfft, dw, t, fmt, rnd, xser == "NEW", 1)
        if C.lib().orc_validate(p, C.FWD if which == "single" else C.PAIR) != 0 or dw + fmt * nfft * (1 if which == "single" else 2) > 62:
            refused += 1
            continue
        ok, whole, n = compare_wrapper(which, nfft, dw, t, fmt, rnd, xser, count=3, seed=rng.randint(1, 1 << 30))
        if not ok:
            print("%s NFFT %d DW %2d TW %2d FORMAT %d RNDMODE %d %s: %d of %d frames, DIFFERENT" % (which, nfft, dw, t, fmt, rnd, xser, whole, n), flush=True)
            bad += 1
        done += 1
    print("rtl_sim --fuzz-wrappers: %d configurations clocked against the oracle, %d more refused by the oracle's validator, %d unexpected, %.0f s"
          % (done, refused, bad, time.time() - t0))
    return 1 if bad else 0


def run_kit(only=None):
    """The external-pin kit (tools/vivado_crosscheck) run on THIS simulation instead of xsim: every case of expected/manifest.json goes through
    the reference's text with the protocol of the kit's testbenches (reset, GAP idle clocks between frames, the enable toggling for the WRAP
    pair, FLUSH all-zero frames, the drain), the dump is written in the testbench's text format and handed to the kit's own compare.py."""
    import json
    import subprocess
    import tempfile
    import time

    import numpy as np

    from intfftk_amd import textio
    kit = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vivado_crosscheck")
    man = json.load(open(os.path.join(kit, "expected", "manifest.json")))
    tmp = tempfile.mkdtemp(prefix="rtl_sim_kit_")
    bad = 0
    print("# tools/rtl_sim.py --kit: the cases of tools/vivado_crosscheck/expected/manifest.json through the reference's own text (what run_xsim.sh does")
    print("# with xsim), judged by the kit's compare.py.  Testbench protocol: single GAP 4 FLUSH 2; pair RAMB_TYPE WRAP GAP 32 FLUSH 6, toggling enable")
    for c in man["cases"]:
        if only and c["case"] not in only:
            continue
        t0 = time.time()
        nfft, dw, t, xser = c["nfft"], c.get("data_width", 16), c.get("twdl_width", 16), c.get("xser", "NEW")
        n = 1 << nfft
        stim = os.path.join(kit, "expected", c["stimulus"])
        hexio = c.get("text") == "hex"
        pair = c["tb"].startswith("tb_pair")
        ow = dw + c["format"] * nfft * (2 if pair else 1)
        if pair:
            x = textio.table_to_double(textio.read_hex(stim, dw), n) if hexio else textio.read_di_double(stim, n)
            rows, _ = run_pair(nfft, dw, t, c["format"], c["rndmode"], xser, x.tolist(), "wrap", gap=32, flush=6, toggle=True, ports=True, lead=32)
        else:
            x = textio.read_hex(stim, dw).reshape(-1, n, 2) if hexio else textio.read_di_single(stim, n)
            rows, _ = run_single_path(nfft, dw, t, c["format"], c["rndmode"], xser, x.tolist(), gap=4, flush=2, lead=16)
        dump = os.path.join(tmp, "%s_%s_rtl.dat" % (c["case"], c["mode"]))
        if hexio:
            textio.write_hex(dump, np.array(rows, dtype=object), ow)
        else:
            np.savetxt(dump, np.array(rows, dtype=np.int64), fmt="%d")
        r = subprocess.run([sys.executable, os.path.join(kit, "compare.py"), c["case"], c["mode"], dump], capture_output=True, text=True)
        print("%s  [%d lines dumped, %.0f s]" % (r.stdout.strip() or r.stderr.strip(), len(rows), time.time() - t0), flush=True)
        bad += r.returncode != 0
    print("rtl_sim --kit: %d cases, %d FAIL" % (len([c for c in man["cases"] if not only or c["case"] in only]), bad))
    return 1 if bad else 0


def main():
    import random
    if not available():
        print("reference not present: nothing to do")
        return 0
    if "--sweep" in sys.argv:
        return sweep()
    if "--fuzz-wrappers" in sys.argv:
        k = sys.argv.index("--fuzz-wrappers")
        return fuzz_wrappers(int(sys.argv[k + 1]), int(sys.argv[k + 2]))
    if "--fuzz" in sys.argv:
        k = sys.argv.index("--fuzz")
        return fuzz(int(sys.argv[k + 1]), int(sys.argv[k + 2]))
    if "--kit" in sys.argv:
        return run_kit(set(sys.argv[sys.argv.index("--kit") + 1:]) or None)
    nfft = int(sys.argv[sys.argv.index("--nfft") + 1]) if "--nfft" in sys.argv else 4
    nfr = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 3
    rng = random.Random(7)
    bad = 0
    for direction in ("FWD", "INV"):
        for (dw, t, fmt, rnd, xser) in ((16, 16, 0, 0, "NEW"), (16, 16, 0, 1, "NEW"), (16, 16, 1, 0, "NEW"), (16, 16, 0, 0, "OLD")):
            frames = [[(rng.randint(-(1 << (dw - 1)), (1 << (dw - 1)) - 1), rng.randint(-(1 << (dw - 1)), (1 << (dw - 1)) - 1)) for _ in range(1 << nfft)]
                      for _ in range(nfr)]
            got, _ = run_core(direction, nfft, dw, t, fmt, rnd, xser, frames)
            want = expected(direction, nfft, dw, t, fmt, rnd, xser, frames)
            ok = got[:len(want)] == want[:len(got)] and len(got) >= len(want) - (1 << (nfft - 1))
            print(direction, "NFFT", nfft, "DW", dw, "TW", t, "FORMAT", fmt, "RNDMODE", rnd, xser, ":", len(got), "beats out,", "equal" if ok else "DIFFERENT", flush=True)
            bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
