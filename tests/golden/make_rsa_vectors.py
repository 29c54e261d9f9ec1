#!/usr/bin/env python3
"""Extracts the reference's own RSA known-answer data -- the (modulus, PKCS#1 v1.5 signature, SHA-256 digest) triples of its
unit tests at src/rsa/chip.rs:706-716 and :751-761 -- into tests/golden/rsa_vectors.json.  Data only (three decimal
integers per vector); run in the build container, where /root/reference exists."""
import json, os, re
HERE = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/src/rsa/chip.rs").read().splitlines()
out = []
for first in (705, 750):                       # 0-based line ranges holding one vector each
    block = "\n".join(src[first:first + 14])
    nums = re.findall(r'from_str\("(\d+)"\)', block)
    assert len(nums) == 3, (first, len(nums))
    n, sign, digest = (int(x) for x in nums)
    out.append({"source": "src/rsa/chip.rs:%d-%d" % (first + 1, first + 14), "n": str(n), "signature": str(sign), "sha256_digest": str(digest), "e": 65537})
json.dump(out, open(os.path.join(HERE, "rsa_vectors.json"), "w"), indent=1)
print("wrote rsa_vectors.json:", [len(v["n"]) for v in out])
