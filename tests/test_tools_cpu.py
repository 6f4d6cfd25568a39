"""Build / measurement tooling that guards results (no GPU): the assembly audit and the staleness rule of the PMC table."""

import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def test_audit_flags_an_inline_vector_instruction_that_reads_a_dot_result():
    import audit_asm_loads as A

    lines = """	v_dot2c_i32_i16_e32 v5, v1, v2
	v_add_u32_e32 v9, v8, v7
	;;#ASMSTART
	v_mad_i32_i16 v6, v5, s2, v3
	;;#ASMEND
	v_dot2c_i32_i16_e32 v6, v1, v2
	v_nop
	v_nop
	v_nop
	v_nop
	;;#ASMSTART
	v_mad_i32_i16 v7, v6, s2, v3
	;;#ASMEND
	;;#ASMSTART
	v_mad_i32_i16 v8, v10, s2, v3
	;;#ASMEND""".split("\n")
    assert A.dot_hazards("k", lines) == 1          # the first one only: the second is outside the window, the third unrelated


def test_audit_flags_a_copy_of_in_flight_registers(tmp_path):
    import audit_asm_loads as A

    asm = """_ZN3wfa11k_sg_runs32ILi11ELi40EEEvNS_8RunsArgsE:
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[2:5], v40, s[4:7], s8 offen
	;;#ASMEND
	v_mov_b32_e32 v50, v3
	;;#ASMSTART
	s_waitcnt vmcnt(5) ; v[2:5]
	;;#ASMEND
	v_mov_b32_e32 v51, v3
.Lfunc_end0:
"""
    path = tmp_path / "k.s"
    path.write_text(asm)
    assert A.main(str(path)) == 1
    path.write_text(asm.replace("	v_mov_b32_e32 v50, v3\n", ""))
    assert A.main(str(path)) == 0


def test_audit_follows_branches(tmp_path):
    """The data-flow form: a register is in flight on ONE of two paths into a block that reads it."""
    import audit_asm_loads as A

    asm = """_ZN3wfa11k_sg_runs32ILi11ELi40EEEvNS_8RunsArgsE:
	s_cbranch_scc1 .LBB0_2
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[128:131], v40, s[4:7], s8 offen
	;;#ASMEND
	s_branch .LBB0_3
.LBB0_2:
	;;#ASMSTART
	buffer_load_dwordx4 v[128:131], v40, s[4:7], s8 offen
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(5) ; v[128:131]
	;;#ASMEND
.LBB0_3:
	v_xor_b32_e32 v1, v2, v129
	s_endpgm
.Lfunc_end0:
"""
    path = tmp_path / "k.s"
    path.write_text(asm)
    assert A.main(str(path)) == 1                      # the fall-through path never waited
    path.write_text(asm.replace("	s_branch .LBB0_3\n", "	;;#ASMSTART\n	s_waitcnt vmcnt(0) ; v[128:131]\n	;;#ASMEND\n	s_branch .LBB0_3\n"))
    assert A.main(str(path)) == 0


def test_traffic_entries_go_stale_with_the_kernel_sources(tmp_path, monkeypatch):
    sys.path.insert(0, REPO)
    import json

    import bench

    key = "k_x|v1725|10|800"
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_digest", lambda repo=None: "aaaaaaaaaaaaaaaa")
    (prof / "hbm_traffic.json").write_text(json.dumps({key: {"bytes": 123, "csrc_sha16": "aaaaaaaaaaaaaaaa"}}))
    assert bench.traffic_entry("k_x", "v1725", 10, 800)[0] == 123
    monkeypatch.setattr(bench, "csrc_digest", lambda repo=None: "bbbbbbbbbbbbbbbb")
    traffic, src = bench.traffic_entry("k_x", "v1725", 10, 800)
    assert traffic is None and "capture again" in src["stale"]
    assert bench.traffic_entry("k_y", "v1725", 10, 800) == (None, None)
