"""Stand-in for `pettingzoo` (absent offline): only the ParallelEnv base-class name.

TEST INFRASTRUCTURE ONLY — see oracle/refstubs/gymnasium/__init__.py.
"""


class ParallelEnv:
    metadata = {}
