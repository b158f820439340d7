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
      "tb_pair_dump": ("tb_pair_dump.vhd", "src/vhdl/main/int_fft_ifft_pair.vhd", "int_fft_ifft_pair")}


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
    for m in re.finditer(r"signal\s+(\w+)\s*:\s*([^;:]+?)(?::=[^;]*)?;", t, re.I):
        w = re.match(r"std_logic_vector\s*\((.*)\s+downto\s+0\s*\)", m.group(2).strip(), re.I)
        sigs[m.group(1).upper()] = "(%s)+1" % w.group(1) if w else None
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
    return {"consts": consts, "signals": sigs, "generic_map": assoc(inst.group(1)), "port_map": assoc(inst.group(2)), "conv_integer": conv}


def ev(expr, env):
    e = re.sub(r"\*\*", "**", expr)
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
        ow = env["DATA_WIDTH"] + env["FORMAT"] * env["NFFT"] * (2 if c["tb"] == "tb_pair_dump" else 1)
        if c.get("out_bits") not in (None, ow):
            errors.append("%s: manifest out_bits %r, entity output width %d" % (tag, c.get("out_bits"), ow))
        import numpy as np

        for fname, bits in ((c["stimulus"], env["DATA_WIDTH"]), (c["expected"], ow)):
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
