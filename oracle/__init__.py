"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's PMF hot path.

Nothing under ``pmf_amd/`` (the product) may import this package.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and there only as the checker / the timed CPU baseline.

Pinning status: the reference (ICEORY/PMF) ships no tests, golden vectors or
KATs for this path.  The oracle is therefore pinned against
outputs of the reference's own modules run in the build container
(``oracle/make_golden.py`` imports them by file path from /root/reference and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).
The camera backbone's arithmetic (torchvision 0.14.1 ``resnet34/50``) is not in
the reference tree and torchvision is not installed here: at that boundary the
restatement follows torchvision's published structure and parity is UNPINNED.
"""
