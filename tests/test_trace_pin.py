"""decision.hpp PINNED against the reference encoder's own loops.

oracle/_ref/turing_ref_trace = the reference encoder with trace points in turing/Search.hpp (oracle/trace_hooks.h, inserted into a temporary copy at
build time): every searchMotionUni (Search.hpp:1315-1355, 2060-2358), searchMotionBi (:1498-1657) and searchIntraPartition (:38-190) of a real encode
writes its inputs (prediction unit, predictors, CABAC rates of mvp_lX_flag, mvPreviousInteger2Nx2N, lambda, most probable modes, rate offsets), every
primitive call it makes with the returned value, and what it decided.  tests/trace_runner.py runs turingcodec_amd/search/decision.hpp -- the text that
is compiled into the product's search kernels -- on those inputs and the encoder's own pictures:

  (also: tu_decision.hpp's decideRqt / decideIntraRd on the encoder's own recorded distortions and RATES -- trace points in Reconstruct.cpp:1296-1428 and
  Search.hpp:143-255 -- must choose the depth / the champion the encoder chose)
  -m "not gpu"  per call over the reference's havoc tables: the same CALL SEQUENCE (positions and values) and the same decisions, for every search of
                the encode, medium / fast / slow, 8- and 10-bit, 1 and 4 encoder threads; + the launch-and-replay batch clients over the stand-in device;
  -m gpu        on the MI355X: the batch clients, and the loops inside the kernels (k_search_list, k_search_bi_list, k_intra_order).
"""
import json
import os
import subprocess
import sys

import pytest

import trace_tools as tt

HERE = os.path.dirname(os.path.abspath(__file__))
needs_trace = pytest.mark.skipif(not tt.have_trace_encoder(), reason="oracle/_ref/turing_ref_trace not built (needs /root/reference; `make -C oracle trace`)")


def run(case, *extra, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(HERE, "trace_runner.py"), case] + list(extra), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def check_cpu(rep, min_searches):
    assert rep["stream_is_the_committed_reference_stream"], "the trace points changed the encode"
    u, b, i = rep["uni_cpu"], rep["bi_cpu"], rep["intra_cpu"]
    assert u["searches"] >= min_searches and u["calls"] > 20 * u["searches"]
    assert u["mismatching_searches"] == 0 and u["mismatching_call_rows"] == 0, u
    assert u["mismatching_in_the_lane_formulation"] == 0, u      # the pattern / raster / sub-sample steps a candidate per "lane" (what k_search_rows runs), emulated on the host
    assert u["previous_2Nx2N_handovers_checked"] > u["searches"] // 2 and u["previous_2Nx2N_handovers_wrong"] == 0, u
    assert b["searches"] >= min_searches // 4 and b["mismatching_searches"] == 0 and b["mismatching_call_rows"] == 0, b
    assert b["mismatching_in_the_lane_formulation"] == 0, b      # the exhaustive grid costed a candidate per "lane" (what k_search_bi runs), emulated on the host
    assert i["partitions"] >= min_searches // 2 and i["mismatching"] == 0, i
    # the reference samples the encoder held for every intra partition (round 5): filter (strong intra smoothing included) and substitution restated == recorded
    nb = rep["intra_neighbours_cpu"]
    assert nb["partitions_with_samples"] >= i["partitions"] * 9 // 10 and nb["filtered_arrays"] > nb["partitions_with_samples"] // 10, nb
    assert nb["filter_mismatching"] == 0 and nb["substitution_mismatching"] == 0 and nb["partitions_with_substituted_samples"] > 0, nb
    if rep["case"] == "ra_medium_qp32":
        assert nb["strong_smoothing_taken"] > 10, nb      # the bi-linear branch is exercised by real content
    # search/amvp.hpp -- the reference's derivation of the two predictors (Mvp.h:195-436) as data-only code -- on the neighbours the encoder's own neighbourPuData() returned
    # and its temporal candidate, for every searchUni call: the predictors the encoder derived
    a = rep["amvp_cpu"]
    assert a["derivations"] == u["searches"] and a["mismatching"] == 0, a
    assert a["with_a_scaled_candidate"] > 0 and a["second_predictor_is_not_zero"] > 0, a      # the scaling and the two-candidate list are exercised
    # search/cand_mode_list.hpp (the function k_intra_gather makes a partition's most probable modes with) on the neighbour modes the encoder's own CandModeList::getCandidate
    # returned, for every searchIntraPartition: the encoder's list and its number of neighbour modes; and the above neighbour counts as DC at the top of a CTU
    cm = rep["cand_mode_list_cpu"]
    assert cm["partitions"] == i["partitions"] and cm["mismatching"] == 0 and cm["neighbours_differ"] > 0 and cm["angular_and_equal"] > 0, cm
    assert cm["above_is_dc_at_the_top_of_a_ctu"] and cm["partitions_at_the_top_of_a_ctu"] > 0, cm
    # picture_order.hpp: neighbourPositionAvailable -- may a prediction unit read a position at all (next CTU row, picture edge, coding order)? -- for the five positions of
    # every searchUni against the three tests of the encoder's neighbourPuData: what the walk on the host and in k_search_rows decides its reads by
    av = rep["availability_cpu"]
    assert av["prediction_units"] == u["searches"] and av["mismatching"] == 0 and min(av["by_position_A0_A1_B0_B1_B2"]) > 0, av
    # tests/merge.hpp -- the reference's merge candidate list (Mvp.h:486-697: spatial candidates with their pruning, the temporal candidate, combined bi-predictive and zero
    # candidates) as data-only code -- on the neighbours the encoder's own PuMergeNeighbour<>::get returned, for every searchMergeModes call: the list the encoder left
    m = rep["merge_cpu"]
    assert m["derivations"] > u["searches"] // 8 and m["mismatching"] == 0, m
    assert m["with_a_pruned_neighbour"] > 0 and m["lists_with_a_bi_predictive_candidate"] > 0, m      # pruning and the combined candidates are exercised
    # ... and the TEMPORAL candidate both lists start from (amvp.hpp: deriveTemporalCandidate, Mvp.h:44-181) on the two cells of the collocated picture the encoder could read:
    # the candidate it derived for every predictor pair and, per list, for every merge list -- the derivations above no longer take anything but motion data as given
    tc = rep["temporal_cpu"]
    assert tc["derivations"] >= a["derivations"] and tc["mismatching"] == 0 and tc["available"] > 0, tc
    # tu_decision.hpp on the encoder's own rates and distortions: the champion of every intra partition's RD refinement, every transform-tree decision
    rd, q = rep["intra_rd_cpu"], rep["rqt_cpu"]
    assert rd["partitions"] == i["partitions"] and rd["rates_measured_by_the_encoder"] > rd["partitions"] and rd["mismatching_champions"] == 0, rd
    assert rd["champion_is_not_the_first_candidate"] > rd["partitions"] // 10, rd      # the decision is not trivial
    assert q["mismatching"] == 0, q
    if rep["case"] == "ra_slow_qp27":      # the residual quadtree is a speed=slow tool (turing/Speed.h: useRqt)
        assert q["decisions_with_a_choice"] > 5000 and 0 < q["chose_split"] < q["decisions_with_a_choice"], q


@needs_trace
@pytest.mark.parametrize("case,min_searches", [("ra_medium_qp32", 3000), ("ra_fast_qp32", 3000), ("ra_medium_10bit_qp27", 2000), ("ra_slow_qp27", 10000),
                                               ("ra_medium_internal10", 1000), ("ra_medium_qp22", 2000)])
def test_restated_loops_make_the_reference_encoders_calls_and_decisions(case, min_searches):
    check_cpu(run(case), min_searches)


@needs_trace
def test_the_same_with_four_encoder_threads(tmp_path):
    """CTU rows on different threads: the trace is followed per thread, the 2Nx2N hand-over in the order the encoder ran the searches"""
    check_cpu(run("ra_medium_qp32", "--threads", "4"), 3000)


@needs_trace
def test_batch_clients_over_the_stand_in_device_decide_what_the_reference_encoder_decided():
    rep = run("ra_medium_qp32", "--device", "mock", "--limit", "500")
    assert rep["uni_device"]["searches"] >= 400 and rep["uni_device"]["mismatching_launch_and_replay"] == 0, rep["uni_device"]
    assert rep["bi_device"]["searches"] >= 400 and rep["bi_device"]["mismatching_launch_and_replay"] == 0, rep["bi_device"]
    assert rep["intra_device"]["partitions"] >= 400 and rep["intra_device"]["mismatching"] == 0, rep["intra_device"]
    g = rep["intra_neighbours_device"]      # the stand-in's gather on the encoder's own samples: copies and filtered copies (strong smoothing included)
    assert g["gathered"] > 1000 and g["gather_unfiltered_mismatching"] == 0 and g["gather_filtered_mismatching"] == 0, g


@needs_trace
@pytest.mark.gpu
@pytest.mark.parametrize("case,min_searches", [("ra_medium_qp32", 3000), ("ra_fast_qp32", 3000), ("ra_medium_10bit_qp27", 2000)])
def test_search_kernels_decide_what_the_reference_encoder_decided(case, min_searches):
    """every motion search / refinement / mode order of a real encode through the MI355X: launch + replay clients AND the loops inside the kernels"""
    rep = run(case, "--device", "real", timeout=1500)
    check_cpu(rep, min_searches)
    u, b, i = rep["uni_device"], rep["bi_device"], rep["intra_device"]
    print(case, json.dumps({"uni": u, "bi": b, "intra": i}))
    assert u["searches"] == rep["uni_cpu"]["searches"] and u["mismatching_launch_and_replay"] == 0 and u["mismatching_loops_in_kernel"] == 0, u
    assert b["searches"] == rep["bi_cpu"]["searches"] and b["mismatching_launch_and_replay"] == 0 and b["mismatching_loops_in_kernel"] == 0, b
    assert i["partitions"] > 0 and i["mismatching"] == 0, i
    g = rep["intra_neighbours_device"]      # k_intra_gather on the encoder's own reference samples
    assert g["gathered"] > 500 and g["gather_unfiltered_mismatching"] == 0 and g["gather_filtered_mismatching"] == 0, g
