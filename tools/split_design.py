#!/usr/bin/env python3
"""One-off helper (round 5): DESIGN.md's sections -> docs/*.md with readable line lengths.  Tables whose cells are paragraphs become
lists of paragraphs (one item per row), every other over-long line is wrapped.  usage: python tools/split_design.py DESIGN.md docs/"""
import os
import re
import sys
import textwrap

WIDTH = 150
NAMES = {"0": None, "1": "boundary.md", "2": "oracle.md", "3": "restructuring.md", "4": "kernels.md", "5": "measurement.md", "6": "multi_gpu.md",
         "7": "scope.md", "8": "tried.md", "9": None, "10": None}


def cells(row):
    parts = [c.strip() for c in row.strip().strip("|").split("|")]
    return parts


def wrap(text, first="", rest=""):
    return textwrap.fill(text, width=WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def convert(lines):
    out, i = [], 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("|"):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            rows = [cells(r) for r in block if not re.match(r"^\|[\s\-|:]+\|?\s*$", r)]
            if max(len(c) for r in rows for c in r) > 240:
                head = rows[0]
                for r in rows[1:]:
                    first = ("- %s" if "**" in r[0] else "- **%s**") % r[0]
                    body = " — ".join("%s: %s" % (h, c) if h and len(head) > 2 else c for h, c in zip(head[1:], r[1:]) if c)
                    out.append(wrap(first + " — " + body, "", "  "))
                out.append("")
            else:
                out.extend(block)
            i = j
            continue
        if len(l) > WIDTH + 30 and not l.startswith("```") and not l.startswith("    "):
            m = re.match(r"^(\s*(?:[-*]|\d+\.)\s+)(.*)$", l)
            if m:
                out.append(wrap(m.group(2), m.group(1), " " * len(m.group(1))))
            else:
                out.append(wrap(l))
        else:
            out.append(l)
        i += 1
    return out


def main(src, dst):
    text = open(src).read().split("\n")
    sections, cur, key = {}, [], "head"
    for l in text:
        m = re.match(r"^## (\d+)\. ", l)
        if m:
            sections[key] = cur
            key, cur = m.group(1), []
        cur.append(l)
    sections[key] = cur
    for k, name in NAMES.items():
        if not name or k not in sections:
            continue
        body = convert(sections[k])
        body[0] = "# " + re.sub(r"^## \d+\. ", "", body[0])
        pre = ["<!-- moved from DESIGN.md section %s in round 5 (tools/split_design.py); `path:line` citations without a repository prefix are" % k,
               "     relative to /root/reference (Fraunhofer-AISEC/rabe 0.4.2); DESIGN.md is the current summary, this file the detail and history -->", ""]
        open(os.path.join(dst, name), "w").write("\n".join(pre + body).rstrip() + "\n")
        print(name, len(body), "lines, longest", max(len(x) for x in body))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
