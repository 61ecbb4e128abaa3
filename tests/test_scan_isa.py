"""The forward scan waits for its hand-issued loads with counted s_waitcnt statements the compiler cannot see;
tools/check_scan_isa.py replays the generated assembly to make sure no instruction touches a destination register
while its load is in flight.  Runs on the CPU (hipcc cross-compiles gfx950)."""
import importlib.util
import os
import shutil
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_scan_isa.py")


def _tool():
    spec = importlib.util.spec_from_file_location("check_scan_isa", TOOL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GOOD = """
k:
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[0:1], off offset:0
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v[0:1], off offset:16
	;;#ASMEND
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_fma_f64 v[20:21], v[10:11], v[12:13], v[20:21]
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[0:1], off offset:32
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_fma_f64 v[20:21], v[14:15], v[16:17], v[20:21]
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v[0:1], off offset:48
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
"""


def test_checker_accepts_a_correct_ring_and_flags_violations():
    t = _tool()
    assert t.check_kernel("k", GOOD.split("\n")) == []
    # a copy of a destination register between the load and its wait
    bad = GOOD.replace("\ts_cbranch_scc1 .LBB0_1", "\tv_mov_b32_e32 v30, v15\n\ts_cbranch_scc1 .LBB0_1")
    assert any("in-flight" in p for p in t.check_kernel("k", bad.split("\n")))
    # a wait that leaves the needed load in flight
    bad = GOOD.replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)", 1)
    assert any("in-flight" in p for p in t.check_kernel("k", bad.split("\n")))
    # spills
    bad = GOOD.replace("\ts_endpgm", "\tscratch_store_dwordx4 off, v[10:13], off\n\ts_endpgm")
    assert "uses scratch memory" in t.check_kernel("k", bad.split("\n"))


# the same ring with the first trip peeled and parked BEHIND the loop (what hipcc does to k_fwd_scan when the kernel
# argument layout shifts its scheduling): in layout order the loop body seems to consume v[10:13] with both loads in flight
PEELED = """
k:
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[0:1], off offset:0
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v[0:1], off offset:16
	;;#ASMEND
	s_branch .LBB0_3
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_fma_f64 v[20:21], v[10:11], v[12:13], v[20:21]
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[0:1], off offset:32
	;;#ASMEND
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_fma_f64 v[20:21], v[14:15], v[16:17], v[20:21]
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v[0:1], off offset:48
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
.LBB0_3:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_fma_f64 v[20:21], v[10:11], v[12:13], v[20:21]
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v[0:1], off offset:32
	;;#ASMEND
	s_branch .LBB0_2
"""


def test_checker_follows_the_control_flow_of_peeled_loops():
    t = _tool()
    assert t.check_kernel("k", PEELED.split("\n")) == []
    # a violation inside the parked block is still found
    bad = PEELED.replace("\ts_branch .LBB0_2", "\tv_mov_b32_e32 v30, v11\n\ts_branch .LBB0_2")
    assert any("in-flight" in p for p in t.check_kernel("k", bad.split("\n")))
    # and so is a missing wait on the path through it
    bad = PEELED.replace(".LBB0_3:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(1)", ".LBB0_3:\n\t;;#ASMSTART\n\ts_waitcnt vmcnt(2)")
    assert any("in-flight" in p for p in t.check_kernel("k", bad.split("\n")))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_forward_scan_kernels_keep_in_flight_registers_untouched():
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok   ") >= 10
    assert r.stdout.count("untracked stores") >= 10  # the store audit of the sweeps that use st_async ran


def test_store_audit_flags_a_store_without_its_nop_and_flat_accesses():
    t = _tool()
    good = ["k:", "\t;;#ASMSTART", "\tglobal_store_dwordx2 v[0:1], v[2:3], off", "\ts_nop 1", "\t;;#ASMEND", "\ts_endpgm"]
    assert t.audit_stores("k", good) == ([], 1, 0)
    bad = good[:3] + ["\t;;#ASMEND", "\tscratch_store_dword off, v2, off", "\ts_endpgm"]
    probs, n_st, n_scr = t.audit_stores("k", bad)
    assert n_st == 1 and n_scr == 1 and any("without its s_nop" in p for p in probs)
    probs, _, _ = t.audit_stores("k", good[:-1] + ["\tflat_load_dword v4, v[0:1]", "\ts_endpgm"])
    assert any("flat_" in p for p in probs)
