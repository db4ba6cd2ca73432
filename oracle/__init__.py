"""CPU oracle for the Stage-1 retrieval hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (semantic-search_amd/) never does.

PARITY UNPINNED: the arithmetic of this path lives in the third-party `faiss`
package (faiss-cpu>=1.7.4, un-vendored, reference requirements.txt:4), which is
not installed here and cannot be; the reference ships no tests, fixtures or
golden vectors for it.  The oracle therefore restates the
*documented* FAISS semantics at the reference's call sites and is checked
against itself (float64 recomputation, brute force on tiny cases) rather than
against reference output.
"""
