"""Hash of the product's native sources (pyft8_amd/csrc/** and include/ft8rx.h): the stored counter profiles under profiles/ carry it,
and bench.py flags a profile as stale when the tree it benchmarks has another one.   python tools/src_hash.py  -> prints the hash"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(root=ROOT):
    h = hashlib.sha256()
    files = [os.path.join(root, "include", "ft8rx.h")]
    for d, _, fs in os.walk(os.path.join(root, "pyft8_amd", "csrc")):
        files += [os.path.join(d, f) for f in fs if f.endswith((".hip", ".h", ".hpp"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
