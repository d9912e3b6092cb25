#!/usr/bin/env python
"""Re-flows the prose of a markdown file to a column limit (default 120): paragraphs and list items are wrapped with their hanging
indent; headings, table rows, code blocks and indented code are left alone.   python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def wrap_file(path, width=120):
    out, para, in_code = [], [], False
    lines = open(path).read().split("\n")

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([*-]|\d+\.)\s+", first)
        if m:
            lead = first[:m.end()]
            body = first[m.end():]
            hang = " " * len(lead)
        else:
            m2 = re.match(r"^(\s*)", first)
            lead = hang = m2.group(1)
            body = first[len(lead):]
        text = " ".join([body] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                                 break_on_hyphens=False) or [lead.rstrip()])
        para.clear()

    for ln in lines:
        if ln.strip().startswith("```"):
            flush()
            in_code = not in_code
            out.append(ln)
            continue
        if in_code or ln.startswith("    ") and not para and not re.match(r"^\s*([*-]|\d+\.)\s", ln):
            flush()
            out.append(ln)
            continue
        if not ln.strip() or ln.startswith("#") or ln.lstrip().startswith("|"):
            flush()
            out.append(ln)
            continue
        if re.match(r"^\s*([*-]|\d+\.)\s+", ln):           # a new list item ends the previous paragraph / item
            flush()
        para.append(ln)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    wrap_file(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120)
