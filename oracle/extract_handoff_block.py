#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Prints the dense-depth branch of CoarseTracker::setCoarseTrackingRef -- the `if (dense_depth != nullptr) { ... }`
statement, tandem/src/FullSystem/CoarseTracker.cpp:654-723 -- from the reference checkout, VERBATIM, for oracle/Makefile.ref to compile
(into oracle/_ref/, git-ignored: reference text is never committed).  One line is ADDED, directly before the statement's closing brace:
`HANDOFF_EXPORT(KRKi, Kt);`, so that the wrapper (oracle/ref_handoff_capi.cpp) can hand the block's own float products to the test."""
import sys

src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if l.strip() == "if (dense_depth != nullptr) {")
depth, end = 0, None
for i in range(start, len(src)):
    depth += src[i].count("{") - src[i].count("}")
    if depth == 0:
        end = i
        break
assert end is not None and 60 < end - start < 80, (start, end)  # the block is 70 lines in the pinned checkout
print("\n".join(src[start:end]))
print("    HANDOFF_EXPORT(KRKi, Kt);")
print(src[end])
