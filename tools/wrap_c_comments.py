#!/usr/bin/env python3
"""Re-wrap the block comments of a C header at 118 columns:  python tools/wrap_c_comments.py include/intfft.h

Inside /* ... */ blocks a paragraph is a run of lines with the same indentation behind the ` * `; a line that starts a new item (an
identifier followed by `(`, `:` or two spaces) starts a new paragraph.  A paragraph is re-wrapped, with its own prefix, only if one
of its lines is too long; a one-line item that is too long takes the deeper-indented lines behind it as its continuation.
Code lines are never touched."""
import re
import sys
import textwrap

W = 118
ITEM = re.compile(r"^(?:[A-Za-z_][A-Za-z0-9_]*(?:\(|  |:)|\d+[.)] |- )")


def wrap(text, first, rest):
    return textwrap.wrap(" ".join(text.split(" ")), W, initial_indent=first, subsequent_indent=rest, break_long_words=False,
                         break_on_hyphens=False)


def flush(par, out):
    """par: list of (prefix, text).  Emit it, re-wrapped if needed."""
    if not par:
        return
    if max(len(p + t) for p, t in par) <= W:
        out += [p + t for p, t in par]
        return
    first = par[0][0]
    rest = par[1][0] if len(par) > 1 else re.sub(r"/\*", " *", first)
    if len(rest) < len(re.sub(r"/\*", " *", first)) and len(par) > 1:
        rest = par[1][0]
    lines = wrap(" ".join(t for _, t in par), first, rest)
    if len(lines) > 1 and lines[-1].strip() in ("*/", "* */"):
        lines[-2] += " */"
        lines.pop()
    out += lines


def main(path):
    src = open(path).read().split("\n")
    out, par, in_c = [], [], False
    for ln in src:
        if not in_c:
            m = re.match(r"^(\s*/\*+ )(\s*)(.*)$", ln)
            if m and "*/" not in ln:
                in_c = True
                par = [(m.group(1) + m.group(2), m.group(3))]
            else:
                out.append(ln)
            continue
        m = re.match(r"^(\s*\* ?)(\s*)(.*)$", ln)
        end = "*/" in ln
        if not m or not m.group(3).strip() or m.group(3).strip() == "*/":
            flush(par, out)
            par = []
            out.append(ln)
            in_c = not end
            continue
        pref, text = m.group(1) + m.group(2), m.group(3)
        cont_pref = re.sub(r"/\*", " *", par[0][0]) if par else None
        same = par and (pref == (par[-1][0] if len(par) > 1 else cont_pref) or
                        (len(par) == 1 and len(par[0][0] + par[0][1]) > W and len(pref) > len(cont_pref) and not ITEM.match(text)))
        if par and (not same or ITEM.match(text)):
            flush(par, out)
            par = []
        par.append((pref, text))
        if end:
            flush(par, out)
            par = []
            in_c = False
    flush(par, out)
    open(path, "w").write("\n".join(out))
    print(path, "longest", max(len(x) for x in out))


if __name__ == "__main__":
    main(sys.argv[1])
