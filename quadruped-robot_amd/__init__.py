"""quadruped-robot_amd: batched convex-MPC + WBC control ticks for quadrupeds on MI355X.

The directory name is not a Python identifier; load it with
    importlib.util.spec_from_file_location("quadruped_robot_amd", ".../quadruped-robot_amd/__init__.py",
                                           submodule_search_locations=[...])
(tests/conftest.py, bench.py and __graft_entry__.py do exactly that via `_load_pkg`).
"""
from . import build as _build          # noqa: F401
from . import workload                  # noqa: F401
from . import shard                     # noqa: F401
from . import rendezvous                # noqa: F401
from . import ticklog                   # noqa: F401
from . import replay                    # noqa: F401
from .qrgpu import (Context, QrgpuError, MissingExtension, lib_path, load_library,   # noqa: F401
                    MPCInterface, WbcLocomotionController, model_desc_struct, status_flags, status_iterations)
from .workload import make_batch, make_batch_sequence, mpc_cfg, model_desc, to_soa, ROBOTS    # noqa: F401
