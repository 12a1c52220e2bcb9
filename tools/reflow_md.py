#!/usr/bin/env python
"""Keeps the hand-written markdown readable in a terminal: wraps paragraphs / list items beyond `--width` characters and turns a table
with a row beyond it into a bullet list ("- **first cell** — second — third", preceded by the header cells in parentheses).
Code fences and tables that fit are left alone.   python tools/reflow_md.py FILE.md [...] [--width 150] [--check]"""
import argparse
import re
import sys
import textwrap


def cells(line):
    return [c.strip() for c in re.split(r"(?<!\\)\|", line.strip().strip("|"))]


def reflow(text, width):
    out, lines, i, fence = [], text.split("\n"), 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            fence = not fence
        if fence:
            out.append(ln); i += 1; continue
        if ln.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= width:
                out += block
            else:
                head = cells(block[0])
                rows = [cells(b) for b in block[1:] if not re.fullmatch(r"\|[\s:|-]+\|?", b.strip())]
                out += textwrap.wrap("(" + " — ".join(head) + ")", width)
                for r in rows:
                    first = r[0] if r[0].startswith(("**", "`")) and " " not in r[0].strip("`*") else r[0]
                    item = "- " + (first if first.startswith("**") else "**" + first + "**" if first else "") + "".join(" — " + c for c in r[1:] if c)
                    out += textwrap.wrap(item, width, subsequent_indent="  ", break_long_words=False, break_on_hyphens=False)
            i = j
            continue
        if len(ln) > width and not ln.startswith("#"):
            m = re.match(r"(\s*)([-*] |\d+\. )?", ln)
            ind = m.group(1) + (" " * len(m.group(2)) if m.group(2) else "")
            out += textwrap.wrap(ln, width, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False)
        else:
            out.append(ln)
        i += 1
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--width", type=int, default=150)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    bad = 0
    for f in a.files:
        text = open(f).read()
        if a.check:
            n = sum(1 for ln in text.split("\n") if len(ln) > a.width + 10)
            if n:
                print(f"{f}: {n} lines beyond {a.width + 10} characters"); bad += 1
            continue
        open(f, "w").write(reflow(text, a.width))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
