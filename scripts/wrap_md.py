#!/usr/bin/env python3
"""Wrap the prose of a Markdown file at <= 160 columns: paragraphs, list items and numbered items are re-flowed with their own indentation; table rows,
headings, code fences and lines that already fit are left alone.  In place."""
import re, sys, textwrap
W = 160
path = sys.argv[1]
out, fence = [], False
for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        fence = not fence
    if fence or len(line) <= W or line.lstrip().startswith("|") or line.startswith("#") or line.lstrip().startswith("{"):
        out.append(line); continue
    m = re.match(r"^(\s*)((?:[-*] |\d+\. )?)", line)
    lead, mark = m.group(1), m.group(2)
    body = line[len(lead) + len(mark):]
    out.extend(textwrap.wrap(body, width=W, initial_indent=lead + mark, subsequent_indent=lead + " " * len(mark), break_long_words=False, break_on_hyphens=False))
open(path, "w").write("\n".join(out))
