"""`import tokengeex` — the import name of the reference's Python package (bindings/python: module
`tokengeex`, classes `Tokenizer` and `TokenGeeXError`, stub bindings/python/tokengeex.pyi:3-255), served by
the MI355X implementation in `tokengeex_amd`: code written against the reference's package runs unchanged.
(The reference creates TokenGeeXError but never adds it to its module, bindings/python/src/lib.rs:226-233;
it is exported here, as its stub promises.)"""
from tokengeex_amd import TokenGeeXError, Tokenizer  # noqa: F401

__all__ = ["Tokenizer", "TokenGeeXError"]
