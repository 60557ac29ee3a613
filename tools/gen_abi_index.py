"""Regenerates the "C ABI index" appendix of INTEGRATION.md from include/tlk.h: every exported entry point, the comment section it
belongs to and the reference locations (file:line) that section cites.  python tools/gen_abi_index.py [--check]"""
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
BEGIN, END = "<!-- abi-index:begin (tools/gen_abi_index.py) -->", "<!-- abi-index:end -->"
CITE = re.compile(r"(?:plugins|tracklab|configs)?/?[\w/]+\.(?:py|yaml):\d+(?:-\d+)?(?:,\d+(?:-\d+)?)*")


def _clean(body):
    body = re.sub(r"^\s*\*\s?", "", body, flags=re.M)
    return re.sub(r"-{10,}", "", body).strip()


def parse(header):
    """Sections are the banner comments (a line of dashes); an entry point also gets the citations of the comment right above it."""
    text = open(header).read()
    out = []
    title, cites = "conventions (header preamble)", []
    last_comment, last_end = "", -1
    for m in re.finditer(r"/\*(.*?)\*/|^(?:const char \*|long long |int )\s*(tlk_\w+)\s*\(", text, re.S | re.M):
        if m.group(2):
            own = []
            if last_end >= 0 and text[last_end:m.start()].strip() == "":        # a comment directly above the declaration
                own = CITE.findall(last_comment.replace("\n", " "))
            out.append((m.group(2), title, sorted(set(cites) | set(own))))
            last_end = -1
        else:
            raw = m.group(1)
            body = _clean(raw)
            if re.search(r"-{10,}", raw) and body:
                title = body.split("\n")[0].strip().rstrip(".")
                cites = sorted(set(CITE.findall(body.replace("\n", " "))))
                last_end = -1
            else:
                last_comment, last_end = body, m.end()
    return out


def table(rows):
    lines = ["| entry point | section of `include/tlk.h` | reference locations cited there |", "|---|---|---|"]
    for name, title, cites in rows:
        t = title if len(title) <= 110 else title[:107] + "..."
        lines.append(f"| `{name}` | {t} | {', '.join('`' + c + '`' for c in cites[:6]) if cites else '--'} |")
    return "\n".join(lines)


def main():
    rows = parse(os.path.join(ROOT, "include", "tlk.h"))
    block = f"{BEGIN}\n{table(rows)}\n{END}"
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    if BEGIN in text:
        new = text[:text.index(BEGIN)] + block + text[text.index(END) + len(END):]
    else:
        new = text.rstrip("\n") + "\n\n## Appendix: C ABI index\n\nGenerated from the header (`python tools/gen_abi_index.py`; `tests/test_abi.py` checks that it is current): every exported\nentry point, the header section that documents it, and the reference code that section names.\n\n" + block + "\n"
    if "--check" in sys.argv:
        sys.exit(0 if new == text else 1)
    open(path, "w").write(new)
    print(f"{len(rows)} entry points")


if __name__ == "__main__":
    main()
