"""Minimal stand-in for the `gymnasium` package (absent offline).

TEST INFRASTRUCTURE ONLY.  Used by oracle/capture_goldens.py, in the build container only, so
that the reference's `src/environment/yard.py` can be imported *unmodified* to record golden
traces.  Nothing in the product path imports this.  Only the 7 names the reference's hot path
touches are provided; the single piece of restated third-party behaviour is
`Graph.from_jsonable` (arrays: nodes/edges in the sub-space dtype, edge_links int32).
"""
from . import spaces  # noqa: F401
