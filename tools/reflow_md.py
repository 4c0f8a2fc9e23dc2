"""Re-wrap the prose of a markdown file to a maximum width (tables, headings, code blocks and indented code stay as they are;
bullets keep their hanging indent).  Usage: python tools/reflow_md.py DESIGN.md 128"""
import re
import sys
import textwrap

path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, para, in_code = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*(?:[*\-+]|\d+\.)\s+)", first)
    if m:
        ind0 = m.group(1)
        ind = " " * len(ind0)
        text = " ".join([first[len(ind0):].strip()] + [l.strip() for l in para[1:]])
        out.extend(textwrap.wrap(text, width=width, initial_indent=ind0, subsequent_indent=ind, break_long_words=False,
                                 break_on_hyphens=False))
    else:
        lead = re.match(r"^\s*", first).group(0)
        text = " ".join(l.strip() for l in para)
        out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=lead, break_long_words=False,
                                 break_on_hyphens=False))
    para = []


for l in lines:
    if l.strip().startswith("```"):
        flush()
        in_code = not in_code
        out.append(l)
        continue
    if in_code or l.startswith("|") or l.startswith("#") or not l.strip():
        flush()
        out.append(l)
        continue
    if re.match(r"^\s*(?:[*\-+]|\d+\.)\s+", l) and para:
        flush()
    para.append(l)
flush()
open(path, "w").write("\n".join(out))
