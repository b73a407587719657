"""Wraps the prose paragraphs of a markdown file at `width` columns (tables, headings, code fences and list structure are
kept; a list item's continuation lines are indented under its text).  python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def wrap(text, width=120):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*(?:[-*+]|\d+[.)])\s+|\s*)", first)
        lead = m.group(1) if m else ""
        body = " ".join([first[len(lead):].strip()] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=" " * len(lead),
                                 break_long_words=False, break_on_hyphens=False) or [lead.rstrip()])
        para.clear()

    for ln in text.split("\n"):
        if ln.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(ln)
        elif fence or ln.startswith("|") or ln.startswith("#") or not ln.strip() or ln.startswith("    "):
            flush()
            out.append(ln)
        elif re.match(r"^\s*(?:[-*+]|\d+[.)])\s+", ln) or ln.startswith("**") and para and para[-1].rstrip().endswith("."):
            flush()
            para.append(ln)
        else:
            para.append(ln)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    src = open(path).read()
    open(path, "w").write(wrap(src, w))
