#!/usr/bin/env python3
"""Re-flow a Markdown file so that nobody has to scroll sideways:  python tools/wrap_md.py FILE... [--width 118] [--check]

* paragraphs and list items are re-wrapped at --width (hanging indent for list items, no breaks inside words);
* a table whose widest row exceeds --width + 8 becomes a list: the first cell of a row is the item, the other cells are
  sub-items labelled with their column heading (a table that fits stays a table);
* headings, fenced code and tables that fit are left alone.
--check only reports lines longer than 160 characters (exit status 1 if there are any).
"""
from __future__ import annotations

import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)([*+-]|\d{1,2}[.)])\s+(?=\S)")


def wrap(text: str, width: int, first: str, rest: str) -> list[str]:
    return textwrap.wrap(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest,
                         break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row: str) -> list[str]:
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|") and not row.endswith("\\|"):
        row = row[:-1]
    out, cur, code = [], "", False
    i = 0
    while i < len(row):
        ch = row[i]
        if ch == "`":
            code = not code
        if ch == "\\" and i + 1 < len(row) and row[i + 1] == "|":
            cur += "|"
            i += 2
            continue
        if ch == "|" and not code:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    out.append(cur.strip())
    return out


def table_to_list(rows: list[str], width: int) -> list[str]:
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        if not any(c):
            continue
        out += wrap(c[0] if c[0] else "(same)", width, "* ", "  ")
        for h, v in zip(head[1:], c[1:]):
            if v:
                out += wrap(("%s: %s" % (h, v)) if h else v, width, "  - ", "    ")
    return out


def reflow(lines: list[str], width: int) -> list[str]:
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i].rstrip("\n")
        if ln.lstrip().startswith("```"):  # fenced code: verbatim
            out.append(ln)
            i += 1
            while i < n and not lines[i].lstrip().startswith("```"):
                out.append(lines[i].rstrip("\n"))
                i += 1
            if i < n:
                out.append(lines[i].rstrip("\n"))
                i += 1
            continue
        if not ln.strip() or ln.startswith("#") or ln.startswith("    ") and not ITEM.match(ln) or ln.startswith("---") or ln.startswith("{"):
            out.append(ln)
            i += 1
            continue
        if ln.startswith("|"):
            j = i
            while j < n and lines[j].startswith("|"):
                j += 1
            rows = [r.rstrip("\n") for r in lines[i:j]]
            if len(rows) >= 2 and re.match(r"^\|[\s:|-]+\|?\s*$", rows[1]) and max(len(r) for r in rows) > width + 8:
                out += table_to_list(rows, width)
            else:
                out += rows
            i = j
            continue
        m = ITEM.match(ln)
        if m:
            first = m.group(0)
            rest = " " * len(first)
            text = ln[len(first):]
            i += 1
            while i < n:
                nx = lines[i].rstrip("\n")
                if not nx.strip() or ITEM.match(nx) or nx.startswith("#") or nx.startswith("|") or nx.lstrip().startswith("```"):
                    break
                text += " " + nx.strip()
                i += 1
            out += wrap(text, width, first, rest)
            continue
        prefix = "> " if ln.startswith(">") else ""
        text = ln[len(prefix):] if prefix else ln
        i += 1
        while i < n:
            nx = lines[i].rstrip("\n")
            if not nx.strip() or ITEM.match(nx) or nx.startswith("#") or nx.startswith("|") or nx.lstrip().startswith("```") or nx.startswith("---"):
                break
            text += " " + (nx[len(prefix):] if prefix and nx.startswith(prefix) else nx).strip()
            i += 1
        out += wrap(text, width, prefix, prefix)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 118
    if "--width" in sys.argv:
        args.remove(str(width))
    bad = 0
    for path in args:
        lines = open(path, encoding="utf-8").read().split("\n")
        if "--check" in sys.argv:
            for k, ln in enumerate(lines, 1):
                if len(ln) > 160:
                    print("%s:%d: %d characters" % (path, k, len(ln)))
                    bad += 1
            continue
        new = reflow(lines, width)
        open(path, "w", encoding="utf-8").write("\n".join(new).rstrip("\n") + "\n")
        print("%s: %d -> %d lines, longest %d" % (path, len(lines), len(new), max(len(x) for x in new)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
