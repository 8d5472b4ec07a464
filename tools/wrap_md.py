"""Re-wrap a Markdown file at <= 118 columns: paragraphs and list items are re-flowed (continuation lines keep the item's indentation),
code fences, tables, headings and HTML are left alone.  usage: python tools/wrap_md.py FILE [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
out, para, in_code = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([-*+] |\d+\. |> )?", first)
    indent = m.group(1) or ""
    bullet = m.group(2) or ""
    body = " ".join([first[len(indent) + len(bullet):].strip()] + [ln.strip() for ln in para[1:]])
    sub = indent + " " * len(bullet)
    out.extend(textwrap.wrap(body, width=width, initial_indent=indent + bullet, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [indent + bullet])
    para = []


for line in open(path).read().split("\n"):
    if line.lstrip().startswith("```"):
        flush()
        in_code = not in_code
        out.append(line)
        continue
    if in_code or line.startswith("|") or line.startswith("#") or line.startswith("<") or not line.strip():
        flush()
        out.append(line)
        continue
    if re.match(r"^\s*([-*+] |\d+\. )", line) or (para and re.match(r"^\s{0,3}\S", line) and re.match(r"^\s+", para[0]) and not line.startswith(" ")):
        flush()
    para.append(line)
flush()
open(path, "w").write("\n".join(out))
