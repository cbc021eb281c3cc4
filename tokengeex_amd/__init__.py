"""tokengeex_amd — MI355X-native (gfx950) implementation of TokenGeeX's Unigram
encode / E-step hot path behind the reference's Tokenizer API.

Importing this package loads tokengeex_amd/libtgx.so (HIP kernels + C ABI); the
import fails if the extension has not been built.  See include/tgx.h for the ABI
and DESIGN.md for the path and its boundary.
"""
from ._lib import (ESTEP_SNIPPET_LEN, MAX_TOKEN_LEN, NativeCorpus, NativeModel, NativeResult,
                   TokenGeeXError, device_count, pack)
from .tokenizer import CrlfProcessor, Tokenizer, UnicodeProcessor, split_special_tokens

__all__ = ["Tokenizer", "TokenGeeXError", "NativeModel", "NativeCorpus", "NativeResult",
           "CrlfProcessor", "UnicodeProcessor", "split_special_tokens", "device_count", "pack",
           "MAX_TOKEN_LEN", "ESTEP_SNIPPET_LEN"]
