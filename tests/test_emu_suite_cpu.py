"""The `gpu` parity tests, run on the CPU against the UNMODIFIED kernel sources (tests/emu: demi_amd/csrc compiled with g++ on
top of a lock-step wave64 emulator - fibers for lanes, a rendezvous for every cross-lane operation, real atomics between
workgroups on different OS threads; the specialised kernels go through a stand-in for hiprtc that compiles the generated
translation unit the same way).  TEST INFRASTRUCTURE: the product never loads any of it; on a GPU box the same tests run on the
GPU (`-m gpu`) and that run is the parity claim.  What this adds without a GPU:

* the logic of K1 / K2 / K3 / k_provenance as written - interpreter and generated code, narrow and wide tables, every K1
  variant - is held against the oracle in the CPU suite too;
* the emulator aborts a launch whose lanes meet at different cross-lane operations (divergent ballot / readlane / shuffle),
  so wave-uniformity of those sites is checked, not assumed;
* `W64_LANE_ORDER=reverse` (or `shuffle[:seed]`: a new permutation per interval) resumes the lanes of every lock-step interval
  in another order than first to last: a kernel whose result
  depended on which lane's stores another lane sees WITHOUT an ordering point in between would change its answers (this is
  how the missing ordering point in k_provenance was found: correct in lock step on the GPU, invisible to the compiler).

The selection below is sized for a couple of minutes; `DEMI_EMU=1 python -m pytest tests -m gpu` runs everything that does not
need a torch device."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_emulated(ids, threads=4, lane_order=None, timeout=1500, extra_env=None, only=None):
    """extra_env: further variables - for the ids listed in `only` (run as a second pytest process), or for all of them."""
    env = dict(os.environ, DEMI_EMU="1", W64_THREADS=str(threads))
    env.pop("DEMI_JIT_DEFINES", None)
    if lane_order:
        env["W64_LANE_ORDER"] = lane_order
    groups = [(ids, dict(env, **(extra_env or {})))]
    if only:
        groups = [([i for i in ids if i not in only], env), ([i for i in ids if i in only], dict(env, **(extra_env or {})))]
    for group, genv in groups:
        if not group:
            continue
        cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x"] + ["tests/" + i for i in group]
        out = subprocess.run(cmd, cwd=ROOT, env=genv, capture_output=True, text=True, timeout=timeout)
        tail = (out.stdout + out.stderr)[-4000:]
        assert out.returncode == 0, tail
        m = re.search(r"(\d+) passed", out.stdout)
        assert m and int(m.group(1)) >= len(group) and "failed" not in out.stdout and "skipped" not in out.stdout, tail


def test_emulator_reports_divergent_cross_lane_operations(tmp_path):
    """The emulator's own check: a ballot under divergent control flow aborts the launch with both source lines."""
    from tests.emu import build
    b = build.build()
    src = tmp_path / "div.cpp"
    src.write_text('#include <hip/hip_runtime.h>\n#include <stdio.h>\n'
                   '__global__ void k(unsigned long long* o) {\n'
                   '  unsigned long long m = 0;\n'
                   '  if (threadIdx.x & 1) m = __ballot(true);\n'
                   '  else m = __ballot(false);\n'
                   '  o[threadIdx.x] = m;\n}\n'
                   '__global__ void ok(unsigned long long* o) { o[threadIdx.x] = __ballot(threadIdx.x & 1) + __shfl((int)threadIdx.x, 5); }\n'
                   'int main(int c, char** v) {\n'
                   '  static unsigned long long o[64];\n'
                   '  if (c > 1) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); return 0; }\n'
                   '  hipLaunchKernelGGL(ok, dim3(1), dim3(64), 0, 0, o);\n'
                   '  printf("%llx %llx\\n", o[0], o[63]);\n  return 0;\n}\n')
    exe = tmp_path / "div"
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-w", "-I" + emu, str(src), "-o", str(exe), "-L" + os.path.dirname(b["runtime"]),
                           "-lw64rt", "-pthread", "-Wl,-rpath," + os.path.dirname(b["runtime"])])
    good = subprocess.run([str(exe)], capture_output=True, text=True)
    assert good.returncode == 0 and good.stdout.split() == ["aaaaaaaaaaaaaaaf", "aaaaaaaaaaaaaaaf"], good.stdout + good.stderr
    bad = subprocess.run([str(exe), "x"], capture_output=True, text=True)
    assert bad.returncode != 0 and "divergent rendezvous" in bad.stderr, bad.stderr


def test_k1_sources_against_the_oracle_on_the_cpu():
    run_emulated(["test_k1_gpu.py::test_raft5_parity_all_capacities[64]",
                  "test_k1_gpu.py::test_first_schedules_against_the_random_scheduler_transliterations_record",
                  "test_k1_gpu.py::test_recorded_event_traces_are_identical",
                  "test_k1_gpu.py::test_random_programs_interpreter_specialised_and_oracle_agree[1]",
                  "test_blocked_actors_gpu.py::test_k1_parity_with_crashed_actors[0]",
                  "test_invariant_gpu.py::test_random_program_invariants_through_every_kernel[1]"])


def test_k1_spread_variant_on_the_cpu():
    """K1's SPREAD variant (few lanes of many wavefronts for launches far smaller than the chip) and the frontier kernels with
    lanes_per_wave < 64, interpreter and compiled table, against the plain launch and the oracle."""
    run_emulated(["test_k1_gpu.py::test_small_launches_on_few_lanes_of_many_waves",
                  "test_random_ddmin_gpu.py"], extra_env={"DEMI_K1_LANES_PER_WAVE": "3"}, only=["test_random_ddmin_gpu.py"])


def test_tables_of_more_than_eight_actors_on_the_cpu():
    """The BIG layout (include/demi_gpu.h): every kernel compiled for the 11-node raft table and the 12-actor shuffle job - K1 and its
    recording / SrcDstFIFO / carried-generator / candidate-frontier variants, K2 with demi_ddmin, K3 in both orders - against the oracle,
    once more with the lanes of every lock-step interval resumed in reverse."""
    run_emulated(["test_big_gpu.py"], timeout=2400)
    run_emulated(["test_big_gpu.py::test_random_scheduler_on_tables_of_more_than_eight_actors", "test_big_gpu.py::test_dpor_in_both_orders"],
                 threads=1, lane_order="reverse", timeout=2400)


def test_k2_sources_against_the_oracle_on_the_cpu():
    run_emulated(["test_k2_gpu.py::test_replay_parity_random_subsequences_raft5",
                  "test_k2_gpu.py::test_bench_candidates_against_the_sts_transliterations_record",
                  "test_k2_gpu.py::test_filter_known_absents_parity[auto]",
                  "test_k2_gpu.py::test_removal_batch_and_kept_parity",
                  "test_k2_gpu.py::test_config4_ddmin_200_external_events",
                  "test_k2_gpu.py::test_native_ddmin_equals_the_python_mirror_on_the_gpu"])


def test_k3_and_provenance_sources_against_the_oracle_on_the_cpu():
    # (one OS thread: the explored-pair table's bounded wait for a half-published key assumes the publishing wave is running)
    run_emulated(["test_k3_gpu.py::test_per_interleaving_outputs_match_the_oracle",
                  "test_k3_gpu.py::test_whole_exploration_identical_gpu_vs_oracle_backend[64]",
                  "test_k3_gpu.py::test_native_exploration_loop_equals_the_python_loop[64-3000]",
                  "test_k3_gpu.py::test_reference_order_on_the_gpu_is_the_batch1_sequence[True]",
                  "test_k3_gpu.py::test_config5_shuffle8_bounded_dpor",
                  "test_k3_gpu.py::test_config5_pipeline_exploration_against_the_oracle",
                  "test_k3_gpu.py::test_one_job_shuffle_in_reference_order_equals_the_scala_transliteration",
                  "test_k3_gpu.py::test_device_queue_and_parent_filter_leave_the_exploration_unchanged",
                  "test_k3_gpu.py::test_reference_order_with_the_record_fetch_in_flight",
                  "test_k3_gpu.py::test_a_round_with_more_points_than_the_staging_area_is_sorted_again",
                  "test_k3_gpu.py::test_config5_pipeline_in_reference_order_is_the_one_at_a_time_sequence",
                  "test_k3_gpu.py::test_reference_order_when_the_speculations_table_is_full",
                  "test_k3_gpu.py::test_checkpointed_interleavings_are_the_same_interleavings",
                  "test_k3_gpu.py::test_device_queue_edge_cases_against_the_host_bookkeeping",
                  "test_provenance_gpu.py"], threads=1)


def test_random_ddmin_and_its_kernel_on_the_cpu():
    """K1 over a frontier of candidate subsequences (a workgroup per candidate) against the oracle per candidate, demi_random_ddmin
    against the reference's DDMin loop around the oracle's RandomScheduler, and its two-rank form (frontier split over the ranks)."""
    run_emulated(["test_random_ddmin_gpu.py"], threads=4)
    run_emulated(["test_comm_gpu.py::test_config4_ddmin_and_config5_dpor_over_the_ranks[2]"], threads=1)


def test_sharded_entry_points_two_ranks_on_the_cpu():
    """demi_random_explore_sharded / demi_replay_batch_sharded / the sharded demi_dpor_explore with two processes, gloo and the
    host all-gather: the worker of tests/test_comm_gpu.py as it is, each rank on an emulated device."""
    run_emulated(["test_comm_gpu.py::test_sharded_entry_points_two_ranks_on_one_gpu"], threads=1)


def test_wide_tables_every_k1_variant_on_the_cpu():
    run_emulated(["test_wide_gpu.py::test_wide_srcdst_fifo_and_carried_generator",
                  "test_wide_gpu.py::test_wide_model_rules_at_the_boundary",
                  "test_wide_gpu.py::test_wide_random_tables_parity[2]"])


def test_array_tables_through_every_kernel_on_the_cpu():
    """DEMI_MODEL_ARRAY (LDX / STX, the replicated-log model): K1 in every variant, K2, the native DDMin, K3."""
    run_emulated(["test_zz_array_gpu.py::test_replicated_log_parity_every_k1_variant[False]",
                  "test_zz_array_gpu.py::test_random_array_tables_parity[2-True-9]",
                  "test_zz_array_gpu.py::test_array_tables_replay_ddmin_and_dpor[True]",
                  "test_zz_array_gpu.py::test_scheduler_mirror_on_a_table_with_arrays",
                  "test_zz_array_gpu.py::test_array_golden_fixtures_on_gpu"])


def test_payload_tables_through_every_kernel_on_the_cpu():
    """DEMI_MODEL_PAYLOADS (LDP / PSET, the 48-bit payload area): the raft with akka-raft's field sets through K1 in every
    variant, recorded traces, K2, the native DDMin and K3; a random table with four 12-bit fields."""
    run_emulated(["test_payloads_gpu.py::test_raft_with_akka_raft_field_sets_through_the_kernels",
                  "test_payloads_gpu.py::test_random_payload_tables_parity[4]"], threads=1)


def test_results_do_not_depend_on_the_order_of_the_lanes_within_an_interval():
    run_emulated(["test_k1_gpu.py::test_raft5_parity_all_capacities[32]",
                  "test_k1_gpu.py::test_srcdst_fifo_parity_raft5[64]",
                  "test_k2_gpu.py::test_replay_parity_random_subsequences_raft5",
                  "test_k2_gpu.py::test_filter_known_absents_parity[hbm]"], lane_order="reverse")
    run_emulated(["test_k3_gpu.py::test_per_interleaving_outputs_match_the_oracle",
                  "test_provenance_gpu.py", "test_zz_array_gpu.py::test_raft_with_a_real_log_through_the_kernels"], threads=1, lane_order="shuffle:7")
