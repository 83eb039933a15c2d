"""CPU oracle: a plain numpy restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``movie-recommender-demo_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / timed baseline, never as the product.

Pinning:
* towers + ranker: pinned against outputs of the reference's own ``two_tower_model.py`` /
  ``transformer_ranker.py`` run in the build container (``tests/golden/make_golden.py``
  -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
* search: faiss (``faiss-cpu>=1.7.4``, requirements.txt:8, not vendored, not installed,
  no golden vectors in the reference) -> **parity unpinned** at the faiss boundary;
  ``IndexFlatIP`` is mathematically exact inner-product top-k, which is what
  ``oracle.search`` restates, anchored on the call sites faiss_retrieval.py:97-166.
"""
from . import towers, ranker, search, pipeline  # noqa: F401
