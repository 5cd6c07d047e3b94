"""Worker for tests/test_host_logic.py::test_two_rank_gloo_matches_single_process (launched by torch.distributed.run)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch.distributed as dist

from epilogos_amd import driver
from tests.fake_backend import OracleBackend

ind, out = Path(sys.argv[1]), Path(sys.argv[2])
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group(backend="gloo")
if (ind / "A").is_dir():                                     # paired layout: in/A/*.txt and in/B/*.txt
    fa, fb = sorted((ind / "A").glob("*")), sorted((ind / "B").glob("*"))
    driver.run_paired_groups(fa, fb, 18, 1, out, "t_s1", 17, -1, 4242, backend=OracleBackend())
else:
    files = sorted(ind.glob("*"))
    driver.run_single_group(files, 18, 1, out, "t_s1", backend=OracleBackend())
if world > 1:
    dist.destroy_process_group()
