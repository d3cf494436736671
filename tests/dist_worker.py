"""Worker for tests/test_bench_dist.py: exercises bench.py's multi-rank plumbing over gloo on CPU."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    rank, local_rank, world = bench.dist_env()
    dist = bench.dist_init(world, local_rank, backend="gloo")
    assert dist is not None and dist.get_world_size() == world
    dist.barrier()
    my_wall = 2.0 + rank            # rank 1 is slower: the job time is the max
    wall = bench.dist_max(dist, my_wall, "cpu")
    value = bench.whole_job_value(1000, 200, 3, world, wall)
    (out_dir / f"rank{rank}.json").write_text(json.dumps({"rank": rank, "seed": bench.seed_for_rank(rank), "wall": wall, "value": value}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
