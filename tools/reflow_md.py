#!/usr/bin/env python3
"""Re-wrap the paragraphs and bullets of a Markdown file at 120 characters (tables, headings and blank lines stay as they are).
DESIGN.md is kept as one front page of lines <= 120 characters (tests/test_docs.py); after editing it: python tools/reflow_md.py DESIGN.md"""
import sys
import textwrap

path = sys.argv[1] if len(sys.argv) > 1 else "DESIGN.md"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
out, par, bullet = [], [], False


def flush():
    global par
    if par:
        text = " ".join(x.strip() for x in par)
        out.extend(textwrap.wrap(text, width=width, subsequent_indent="  " if bullet else "", break_long_words=False, break_on_hyphens=False))
        par = []


for line in open(path, encoding="utf-8").read().split("\n"):
    if line.startswith("|") or line.startswith("#") or line.startswith("```") or not line.strip():
        flush()
        out.append(line)
        continue
    if line.startswith("* "):
        flush()
        par, bullet = [line], True
        continue
    if par and bullet and not line.startswith("  "):
        flush()
        bullet = False
    if not par:
        bullet = line.startswith("* ")
    par.append(line)
flush()
open(path, "w", encoding="utf-8").write("\n".join(out))
