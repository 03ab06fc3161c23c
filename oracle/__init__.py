"""CPU oracle for the exact-similarity-search + gathered-InfoNCE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vllm_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

It restates, on the CPU with torch/numpy, the reference functions listed in
SURVEY.md section 8(a).  Every function cites the reference file:line it
follows.  Parity pinning: the reference has no tests or golden vectors of its
own, so the oracle is pinned against outputs of the reference itself run in the
build container: ``oracle/gen_golden.py`` imports ``model/abc_util.py`` and
``visualization/neg_mine.py`` from /root/reference, compiles the hot-path function
definitions of ``evaluate/eval_mscoco.py``, ``evaluate/eval_flickr30k.py`` and
``pretrain/create_negative_index.py`` out of those files (their modules cannot be
imported there: ordinary ImportError/SyntaxError), runs
them on CPU and writes small fixtures to ``tests/golden/``; the ``Linear`` / ``MLP`` /
``Temperature`` classes of ``model/modeling_abc.py`` (which needs ``timm``) are pinned the
same way (``head.npz``).
"""
from .search_oracle import *  # noqa: F401,F403
from .loss_oracle import *  # noqa: F401,F403
