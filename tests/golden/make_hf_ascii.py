"""Generates tests/golden/hf_ascii.json: ASCII golden vectors from HuggingFace
`tokenizers` Unigram — the upstream the reference's model.rs/lattice.rs/trie.rs
were forked from (see their file headers) — as an INDEPENDENT cross-check of the
oracle.  It is comparable only for ASCII text over an ASCII vocab that contains
every used byte (HF advances by chars and falls back to <unk>; TokenGeeX works on
bytes).  Run:  python tests/golden/make_hf_ascii.py
"""
import json
import os
import random

from tokenizers import Tokenizer
from tokenizers.models import Unigram

random.seed(20240607)
ALPHA = [chr(c) for c in range(32, 127)] + ["\n", "\t"]
WORDS = ["the", "def", "return", "self", "import", "for", "in", "if", "else", "class", "int", "str",
         "print", "len", "range", "None", "True", "False", "value", "key", "data", "index", "token"]


def make_vocab(n_multi):
    vocab = {}
    for ch in ALPHA:
        vocab[ch] = -random.uniform(6.0, 9.0)
    pieces = set()
    while len(pieces) < n_multi:
        r = random.random()
        if r < 0.3:
            w = random.choice(WORDS)
            a = random.randrange(0, len(w))
            b = random.randrange(a + 1, len(w) + 1)
            p = w[a:b]
            if random.random() < 0.3:
                p = " " + p
        elif r < 0.5:
            p = random.choice([" ", "\n", "\t"]) * random.randint(2, 8)
        else:
            p = "".join(random.choice("abcdefghij _=().") for _ in range(random.randint(2, 6)))
        if len(p) >= 2:
            pieces.add(p)
    for p in sorted(pieces):
        # a handful of exact ties on purpose (quantised scores)
        vocab[p] = -round(random.uniform(3.0, 12.0), 1 if random.random() < 0.3 else 6)
    items = list(vocab.items())
    random.shuffle(items)
    return items


def make_text(n):
    out = []
    while sum(len(x) for x in out) < n:
        r = random.random()
        if r < 0.5:
            out.append(random.choice(WORDS))
        elif r < 0.7:
            out.append(random.choice([" ", "\n", "    ", "\t", "  "]))
        else:
            out.append("".join(random.choice("abcdefghij _=().") for _ in range(random.randint(1, 9))))
    return "".join(out)[:n]


cases = []
for n_multi, lens in [(40, [0, 1, 2, 7, 64, 300]), (400, [5, 130, 1000, 4000]), (1500, [2500])]:
    vocab = make_vocab(n_multi)
    tk = Tokenizer(Unigram(vocab, None, False))
    texts = [make_text(n) for n in lens]
    texts += ["aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", "                 ", "abababababababababab", "\n\n\n\n\n\n\n"]
    ids = [tk.encode(t).ids for t in texts]
    cases.append({"vocab": vocab, "texts": texts, "ids": ids})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_ascii.json")
with open(out, "w", encoding="utf-8") as f:
    json.dump({"generator": "tests/golden/make_hf_ascii.py", "tokenizers_version": __import__("tokenizers").__version__,
               "cases": cases}, f, ensure_ascii=True)
print(out, os.path.getsize(out), "bytes", sum(len(c["texts"]) for c in cases), "texts")
