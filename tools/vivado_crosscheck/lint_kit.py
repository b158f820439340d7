#!/usr/bin/env python3
"""Static lint of the external-pin kit against the reference's entity declarations (no VHDL front end exists in the build image,
so this is what can be checked before the kit meets a simulator):

  * every generic / port the kit testbenches associate exists, with that spelling, in the reference entity
    (src/vhdl/main/int_fft_single_path.vhd:85-113, int_fft_ifft_pair.vhd:74-107), and every entity port is associated;
  * for every case of expected/manifest.json, each signal bound to a port has the port's width (both width expressions are
    evaluated with the case's generics), and no `conv_integer` operand is wider than 32 bits (a VHDL integer);
  * stimulus values fit DATA_WIDTH and the committed expectations fit the output width.

    python tools/vivado_crosscheck/lint_kit.py [--reference /root/reference]     exit 0 = clean
"""
import argparse
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TB = {"tb_single_dump": ("tb_single_dump.vhd", "src/vhdl/main/int_fft_single_path.vhd", "int_fft_single_path"),
      "tb_pair_dump": ("tb_pair_dump.vhd", "src/vhdl/main/int_fft_ifft_pair.vhd", "int_fft_ifft_pair"),
      # round 6: hexadecimal full-width text I/O (hread / hwrite of ieee.std_logic_textio) for widths beyond a VHDL integer
      "tb_single_hex": ("tb_single_hex.vhd", "src/vhdl/main/int_fft_single_path.vhd", "int_fft_single_path"),
      "tb_pair_hex": ("tb_pair_hex.vhd", "src/vhdl/main/int_fft_ifft_pair.vhd", "int_fft_ifft_pair")}
PAIR_TBS = ("tb_pair_dump", "tb_pair_hex")


def strip_comments(text):
    return "\n".join(line.split("--")[0] for line in text.splitlines())


def entity_decl(text, name):
    """{'generics': {NAME: default}, 'ports': {NAME: (dir, width expression or None for std_logic)}} of `entity name is ... end`"""
    t = strip_comments(text)
    m = re.search(r"entity\s+%s\s+is(.*?)end\s+%s" % (name, name), t, re.S | re.I)
    if not m:
        raise SystemExit("entity %s not found" % name)
    body = m.group(1)
    g = re.search(r"generic\s*\((.*?)\)\s*;\s*port", body, re.S | re.I)
    p = re.search(r"port\s*\((.*)\)\s*;", body, re.S | re.I)
    generics, ports = {}, {}
    for item in g.group(1).split(";"):
        mm = re.match(r"\s*(\w+)\s*:\s*(\w+)\s*(?::=\s*(.*))?$", item.strip(), re.S)
        if mm:
            generics[mm.group(1).upper()] = (mm.group(3) or "").strip()
    for item in p.group(1).split(";"):
        mm = re.match(r"\s*(\w+)\s*:\s*(in|out|inout)\s+(.*)$", item.strip(), re.S | re.I)
        if not mm:
            continue
        typ = mm.group(3).strip()
        w = re.match(r"std_logic_vector\s*\((.*)\s+downto\s+0\s*\)", typ, re.I)
        ports[mm.group(1).upper()] = (mm.group(2).lower(), "(%s)+1" % w.group(1) if w else None)
    return {"generics": generics, "ports": ports}


def tb_decl(text, uut_entity):
    t = strip_comments(text)
    consts = {m.group(1).upper(): m.group(2).strip() for m in re.finditer(r"constant\s+(\w+)\s*:\s*integer\s*:=\s*([^;]+);", t, re.I)}
    sigs = {}
    for m in re.finditer(r"signal\s+([\w\s,]+?)\s*:\s*([^;:]+?)(?::=[^;]*)?;", t, re.I):
        w = re.match(r"std_logic_vector\s*\((.*)\s+downto\s+0\s*\)", m.group(2).strip(), re.I)
        for name in m.group(1).split(","):  # (declaration lists: "signal d0_re, d1_re, d0_im, d1_im : ...")
            sigs[name.strip().upper()] = "(%s)+1" % w.group(1) if w else None
    inst = re.search(r"entity\s+work\.%s\s+generic\s+map\s*\((.*?)\)\s*port\s+map\s*\((.*?)\)\s*;" % uut_entity, t, re.S | re.I)
    if not inst:
        raise SystemExit("instantiation of %s not found" % uut_entity)

    def assoc(s):
        out = {}
        for item in s.split(","):
            a, b = item.split("=>")
            out[a.strip().upper()] = b.strip()
        return out

    conv = [m.group(1).strip().upper() for m in re.finditer(r"conv_integer\s*\(\s*(\w+)\s*\)", t, re.I)]
    # hex text I/O: variables (std_logic_vector) handed to hread / hwrite, slices of hread variables driven onto signals, SXT(signal, SIZE)
    variables = {}
    for m in re.finditer(r"variable\s+([\w\s,]+?)\s*:\s*std_logic_vector\s*\((.*?)\s+downto\s+0\s*\)", t, re.I):
        for v in m.group(1).split(","):
            variables[v.strip().upper()] = "(%s)+1" % m.group(2)
    hread = [m.group(1).upper() for m in re.finditer(r"hread\s*\(\s*\w+\s*,\s*(\w+)\s*\)", t, re.I)]
    hwrite = [m.group(1).upper() for m in re.finditer(r"hwrite\s*\(\s*\w+\s*,\s*(\w+)\s*\)", t, re.I)]
    slices = [(m.group(1).upper(), m.group(2).upper(), "(%s)+1" % m.group(3))
              for m in re.finditer(r"(\w+)\s*<=\s*(\w+)\s*\(\s*(.*?)\s+downto\s+0\s*\)\s*;", t, re.I)]
    sxt = [(m.group(1).upper(), m.group(2).upper(), m.group(3).strip())
           for m in re.finditer(r"(\w+)\s*:=\s*SXT\s*\(\s*(\w+)\s*,\s*([^)]+)\)", t, re.I)]
    uses_textio = re.search(r"use\s+ieee\.std_logic_textio\.all", t, re.I) is not None
    return {"consts": consts, "signals": sigs, "generic_map": assoc(inst.group(1)), "port_map": assoc(inst.group(2)), "conv_integer": conv,
            "variables": variables, "hread": hread, "hwrite": hwrite, "slices": slices, "sxt": sxt, "uses_textio": uses_textio}


def read_hex_words(path):
    """[(value, digits)] rows of a hex text file: two's complement of each word's own digit count"""
    rows = []
    for line in open(path):
        f = line.split()
        if f:
            rows.append([((int(w, 16) - (1 << (4 * len(w)))) if int(w, 16) >> (4 * len(w) - 1) else int(w, 16), len(w)) for w in f])
    return rows


def ev(expr, env):
    e = re.sub(r"(?<!/)/(?!/)", "//", expr)  # VHDL integer division
    return int(eval(e, {"__builtins__": {}}, {k: v for k, v in env.items()}))  # noqa: S307 -- arithmetic on integers from our own files


def lint(reference, kit=HERE):
    errors = []
    manifest = json.load(open(os.path.join(HERE, "expected", "manifest.json")))
    decl = {}
    for tb, (tbfile, reffile, ent) in TB.items():
        e = entity_decl(open(os.path.join(reference, reffile), encoding="latin-1").read(), ent)
        k = tb_decl(open(os.path.join(kit, tbfile)).read(), ent)
        decl[tb] = (e, k)
        for f in k["generic_map"]:
            if f not in e["generics"]:
                errors.append("%s: generic %s is not a generic of %s" % (tbfile, f, ent))
        for f in k["port_map"]:
            if f not in e["ports"]:
                errors.append("%s: port %s is not a port of %s" % (tbfile, f, ent))
        for f in e["ports"]:
            if f not in k["port_map"]:
                errors.append("%s: port %s of %s is left unassociated" % (tbfile, f, ent))
    for c in manifest["cases"]:
        e, k = decl[c["tb"]]
        env = {"NFFT": c["nfft"], "DATA_WIDTH": c.get("data_width", 16), "TWDL_WIDTH": c.get("twdl_width", 16), "FORMAT": c["format"],
               "RNDMODE": c["rndmode"]}
        for name, expr in k["consts"].items():
            try:
                env[name] = ev(expr.upper(), env)
            except Exception:  # noqa: BLE001 -- non-integer constants are irrelevant here
                pass
        tag = "%s/%s" % (c["case"], c["mode"])
        for formal, actual in k["port_map"].items():
            a = actual.upper()
            if a not in k["signals"] or formal not in e["ports"]:
                continue  # literal ('1'), or a formal already reported as unknown
            pw, sw = e["ports"][formal][1], k["signals"][a]
            if (pw is None) != (sw is None):
                errors.append("%s: %s <= %s: std_logic against a vector" % (tag, formal, actual))
            elif pw is not None and ev(pw.upper(), env) != ev(sw.upper(), env):
                errors.append("%s: %s is %d bits, signal %s is %d" % (tag, formal, ev(pw.upper(), env), actual, ev(sw.upper(), env)))
        for s in k["conv_integer"]:
            w = k["signals"].get(s)
            if w is not None and ev(w.upper(), env) > 32:
                errors.append("%s: conv_integer(%s) on %d bits overflows a VHDL integer" % (tag, s, ev(w.upper(), env)))
        if c.get("text") == "hex":  # hread / hwrite operands: whole hex digits, wide enough for the port they feed / dump
            if not k["uses_textio"]:
                errors.append("%s: %s uses hread / hwrite without ieee.std_logic_textio" % (tag, TB[c["tb"]][0]))
            for v in k["hread"] + k["hwrite"]:
                w = k["variables"].get(v)
                if w is None:
                    errors.append("%s: %s is handed to hread / hwrite but is not a std_logic_vector variable" % (tag, v))
                elif ev(w.upper(), env) % 4:
                    errors.append("%s: hread / hwrite operand %s is %d bits: not a whole number of hex digits" % (tag, v, ev(w.upper(), env)))
            for sig, var, w in k["slices"]:  # signal <= variable(W-1 downto 0)
                if var in k["variables"] and sig in k["signals"] and k["signals"][sig] is not None:
                    sw, vw, cut = ev(k["signals"][sig].upper(), env), ev(k["variables"][var].upper(), env), ev(w.upper(), env)
                    if cut != sw:
                        errors.append("%s: %s is %d bits but takes a %d-bit slice of %s" % (tag, sig, sw, cut, var))
                    if vw < cut:
                        errors.append("%s: hread variable %s (%d bits) is narrower than the %d bits taken from it" % (tag, var, vw, cut))
            for var, sig, size in k["sxt"]:  # variable := SXT(signal, SIZE)
                if var in k["variables"] and sig in k["signals"] and k["signals"][sig] is not None:
                    vw, sw, sz = ev(k["variables"][var].upper(), env), ev(k["signals"][sig].upper(), env), ev(size.upper(), env)
                    if sz != vw or sz < sw:
                        errors.append("%s: %s := SXT(%s, %d): variable is %d bits, signal %d" % (tag, var, sig, sz, vw, sw))
        elif c.get("data_width", 16) > 32 or (c.get("out_bits") or 0) > 32:
            errors.append("%s: widths beyond 32 bits need the hex testbenches (a VHDL integer holds 32 bits)" % tag)
        ow = env["DATA_WIDTH"] + env["FORMAT"] * env["NFFT"] * (2 if c["tb"] in PAIR_TBS else 1)
        if c.get("out_bits") not in (None, ow):
            errors.append("%s: manifest out_bits %r, entity output width %d" % (tag, c.get("out_bits"), ow))
        import numpy as np

        for fname, bits in ((c["stimulus"], env["DATA_WIDTH"]), (c["expected"], ow)):
            if c.get("text") == "hex":
                rows = read_hex_words(os.path.join(HERE, "expected", fname))
                cols = 4 if c["tb"] in PAIR_TBS else 2
                if any(len(r) != cols for r in rows):
                    errors.append("%s: %s does not hold %d words per line" % (tag, fname, cols))
                if any(d != (bits + 3) // 4 for r in rows for _, d in r):
                    errors.append("%s: %s holds words that are not %d hex digits (4 * ceil(%d / 4) bits)" % (tag, fname, (bits + 3) // 4, bits))
                if any(v < -(1 << (bits - 1)) or v >= (1 << (bits - 1)) for r in rows for v, _ in r):
                    errors.append("%s: %s holds values outside %d bits" % (tag, fname, bits))
                continue
            a = np.loadtxt(os.path.join(HERE, "expected", fname), dtype=np.int64, ndmin=2)
            if a.min() < -(1 << (bits - 1)) or a.max() >= (1 << (bits - 1)):
                errors.append("%s: %s holds values outside %d bits" % (tag, fname, bits))
    return errors, manifest


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    errs, man = lint(a.reference)
    for e_ in errs:
        print("LINT:", e_)
    print("%d cases, %d problems" % (len(man["cases"]), len(errs)))
    sys.exit(1 if errs else 0)
