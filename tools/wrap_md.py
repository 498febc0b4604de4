"""Wrap a Markdown file at 120 columns (round-5 review, item 9): paragraphs and list items are re-flowed; a table that has a cell
longer than CELL_MAX characters is rewritten as a bulleted list (one bullet per row, `header: cell` parts), because a table row
cannot be wrapped; code fences, headings and narrow tables are left alone.   python tools/wrap_md.py DESIGN.md [--check]"""
import re
import sys
import textwrap

W, CELL_MAX = 120, 110
ITEM = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")


def wrap_block(lines):
    first = lines[0]
    m = ITEM.match(first)
    if m:
        indent = " " * len(m.group(1))
        lead = indent + m.group(2) + " "
        body = ITEM.sub("", first, count=1)
        sub = indent + " " * (len(m.group(2)) + 1)
    else:
        indent = re.match(r"^(\s*)", first).group(1)
        lead, sub, body = indent, indent, first.strip()
    text = " ".join([body.strip()] + [l.strip() for l in lines[1:]])
    return textwrap.wrap(text, width=W, initial_indent=lead, subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [lead.rstrip()]


def table_to_list(rows):
    cells = [[c.strip() for c in r.strip().strip("|").split("|")] for r in rows]
    # (cells containing escaped pipes or code with pipes are rare here; rows with a different cell count are kept by joining the tail)
    hdr = cells[0]
    out = []
    for r in cells[2:]:
        if len(r) > len(hdr):
            r = r[:len(hdr) - 1] + [" | ".join(r[len(hdr) - 1:])]
        parts = []
        for h, c in zip(hdr[1:], r[1:]):
            if c and c not in ("-", "—"):
                parts.append(("*%s*: %s" % (h, c)) if h else c)
        head = r[0] if r and r[0] else "(cont.)"
        text = "* **%s** - %s" % (head.strip("*"), "; ".join(parts)) if parts else "* **%s**" % head.strip("*")
        out += textwrap.wrap(text, width=W, subsequent_indent="  ", break_long_words=False, break_on_hyphens=False)
    return out


def process(src):
    out, i, lines = [], 0, src.split("\n")
    in_code = False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            in_code = not in_code
            out.append(ln); i += 1; continue
        if in_code or not ln.strip() or ln.startswith("#") or ln.startswith("{"):
            out.append(ln); i += 1; continue
        if ln.lstrip().startswith("|"):
            j = i
            while j < len(lines) and lines[j].lstrip().startswith("|"):
                j += 1
            rows = lines[i:j]
            wide = any(len(c) > CELL_MAX for r in rows for c in r.strip().strip("|").split("|")) or any(len(r) > 2 * W for r in rows)
            out += table_to_list(rows) if (wide and len(rows) >= 3) else rows
            i = j; continue
        # a paragraph or list item: up to the next blank line / item / table / heading / fence
        j = i + 1
        while j < len(lines) and lines[j].strip() and not ITEM.match(lines[j]) and not lines[j].lstrip().startswith("|") \
                and not lines[j].startswith("#") and not lines[j].lstrip().startswith("```"):
            j += 1
        out += wrap_block(lines[i:j])
        i = j
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    src = open(path).read()
    res = process(src)
    if "--check" in sys.argv:
        long_ = [k + 1 for k, l in enumerate(res.split("\n")) if len(l) > W]
        print("%d lines, %d longer than %d columns (first: %s)" % (res.count("\n") + 1, len(long_), W, long_[:10]))
    else:
        open(path, "w").write(res)
