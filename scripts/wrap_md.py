#!/usr/bin/env python3
"""Re-flow the prose of a Markdown file at <= 160 columns.  Consecutive prose lines are ONE block (what Markdown renders them as): a paragraph, or a list item
with its continuation lines; a block is joined and wrapped with its own indentation.  Table rows, headings, code fences, blank lines and JSON lines are left alone and end a
block; a line that starts with `**` or a list marker starts a new one.  Blocks whose lines all fit are left as they are.  In place; the sequence of words never changes."""
import re, sys, textwrap
W = 160
ITEM = re.compile(r"^(\s*)([-*] |\d+\. )")
path = sys.argv[1]
lines = open(path).read().split("\n")
blocks, cur, fence = [], None, False          # a block: [first-line indent + marker, continuation indent, [texts], [original lines]]


def close():
    global cur
    if cur is not None: blocks.append(cur); cur = None


for line in lines:
    if line.lstrip().startswith("```"):
        close(); fence = not fence; blocks.append(line); continue
    s = line.lstrip()
    if fence or not s or s.startswith("|") or line.startswith("#") or s.startswith("{") or s.startswith("<"):
        close(); blocks.append(line); continue
    m = ITEM.match(line)
    if m:
        close(); lead = m.group(1) + m.group(2); cur = [lead, " " * len(lead), [line[len(lead):].strip()], [line]]
    elif cur is None or s.startswith("**"):
        close(); lead = line[:len(line) - len(s)]; cur = [lead, lead, [s.strip()], [line]]
    else:
        cur[2].append(s.strip()); cur[3].append(line)
close()
out = []
for b in blocks:
    if isinstance(b, str): out.append(b); continue
    first, cont, texts, orig = b
    if all(len(l) <= W for l in orig): out.extend(orig); continue
    out.extend(textwrap.wrap(" ".join(texts), width=W, initial_indent=first, subsequent_indent=cont, break_long_words=False, break_on_hyphens=False))
open(path, "w").write("\n".join(out))
