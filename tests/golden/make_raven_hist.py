"""Writes tests/golden/raven_byte_hist.json: the byte histogram of the reference's own training
text, /root/reference/books/the_raven.txt (public-domain Project Gutenberg text; SURVEY.md
section 2 row 22, section 8d "char-grid input layers").  Data only: 128 counts of the ASCII
bytes (the reference drops non-ASCII characters: single_text_dataset.py:102-123 encodes with
`ascii`, errors ignored) -- no text of the book is copied.  Run in the build container, where
/root/reference exists; bench.py and the tests read the JSON, never the book.

    python tests/golden/make_raven_hist.py
"""
import json
import os

SRC = "/root/reference/books/the_raven.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raven_byte_hist.json")


def main():
    raw = open(SRC, "rb").read()
    # as the reference reads it: text mode (universal newlines), `[0:max_size]` with the default
    # max_size = -1 (drops the last character), then encode("ascii", "ignore")
    # (single_text_dataset.py:234-244, :112)
    with open(SRC, "r", encoding="utf-8") as f:
        text = f.read()[0:-1].encode("ascii", "ignore")
    counts = [0] * 128
    for b in text:
        counts[b] += 1
    out = {
        "source": "books/the_raven.txt of jloveric/language-interpolation (Project Gutenberg text)",
        "file_bytes": len(raw),
        "ascii_chars": len(text),
        "distinct": sum(1 for c in counts if c),
        "min_byte": min(i for i, c in enumerate(counts) if c),
        "max_byte": max(i for i, c in enumerate(counts) if c),
        "counts": counts,
    }
    with open(DST, "w") as f:
        json.dump(out, f)
    print({k: v for k, v in out.items() if k != "counts"})


if __name__ == "__main__":
    main()
