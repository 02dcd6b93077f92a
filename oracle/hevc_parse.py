"""oracle/hevc_parse.py -- TEST INFRASTRUCTURE.  Minimal HEVC Annex-B / parameter-set / slice-header parser (ITU-T H.265
7.3.1-7.3.6, the order TEncCavlc.cpp writes) used to inspect the reference's fixture bitstreams and to check the
product's bitstream writer field by field."""
import re


def split_annexb(b):
    """-> list of (start_code_len, nal_bytes)."""
    out = []
    pos = [m.start() for m in re.finditer(b"\x00\x00\x01", b)]
    for k, p in enumerate(pos):
        end = pos[k + 1] if k + 1 < len(pos) else len(b)
        sc = 3
        if p > 0 and b[p - 1] == 0:
            sc = 4
        if k + 1 < len(pos) and end > 0 and b[end - 1] == 0:
            end -= 1
        out.append((sc, b[p + 3:end]))
    return out


def unescape(nal):
    return re.sub(b"\x00\x00\x03", b"\x00\x00", nal)


class Bits:
    def __init__(self, data):
        self.d, self.p = data, 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def left(self):
        return len(self.d) * 8 - self.p


def ptl(r, f):
    f["profile_space"] = r.u(2); f["tier"] = r.u(1); f["profile_idc"] = r.u(5)
    f["compat"] = r.u(32)
    f["progressive"] = r.u(1); f["interlaced"] = r.u(1); f["non_packed"] = r.u(1); f["frame_only"] = r.u(1)
    f["reserved43"] = r.u(43); f["inbld"] = r.u(1)
    f["level_idc"] = r.u(8)


def parse_vps(n):
    r = Bits(unescape(n)[2:]); f = {}
    f["vps_id"] = r.u(4); f["base_internal"] = r.u(1); f["base_avail"] = r.u(1); f["max_layers_m1"] = r.u(6); f["max_sub_m1"] = r.u(3)
    f["nesting"] = r.u(1); f["ffff"] = r.u(16); ptl(r, f)
    f["sub_layer_ordering"] = r.u(1); f["max_dec_m1"] = r.ue(); f["num_reorder"] = r.ue(); f["max_latency_p1"] = r.ue()
    f["max_layer_id"] = r.u(6); f["num_layer_sets_m1"] = r.ue(); f["timing"] = r.u(1)
    assert f["timing"] == 0
    f["ext"] = r.u(1); f["trailing"] = (r.u(r.left()), )
    return f


def parse_sps(n):
    r = Bits(unescape(n)[2:]); f = {}
    f["vps_id"] = r.u(4); f["max_sub_m1"] = r.u(3); f["nesting"] = r.u(1); ptl(r, f)
    f["sps_id"] = r.ue(); f["chroma_format"] = r.ue(); f["width"] = r.ue(); f["height"] = r.ue(); f["conf_win"] = r.u(1)
    if f["conf_win"]:
        f["conf"] = [r.ue() for _ in range(4)]
    f["bd_luma_m8"] = r.ue(); f["bd_chroma_m8"] = r.ue(); f["log2_poc_m4"] = r.ue(); f["sub_layer_ordering"] = r.u(1)
    f["max_dec_m1"] = r.ue(); f["num_reorder"] = r.ue(); f["max_latency_p1"] = r.ue()
    f["log2_min_cb_m3"] = r.ue(); f["log2_diff_cb"] = r.ue(); f["log2_min_tb_m2"] = r.ue(); f["log2_diff_tb"] = r.ue()
    f["tu_depth_inter"] = r.ue(); f["tu_depth_intra"] = r.ue(); f["scaling_list"] = r.u(1); assert not f["scaling_list"]
    f["amp"] = r.u(1); f["sao"] = r.u(1); f["pcm"] = r.u(1); assert not f["pcm"]
    f["num_st_rps"] = r.ue(); f["st_rps"] = []
    for i in range(f["num_st_rps"]):
        inter = r.u(1) if i > 0 else 0
        assert not inter
        nneg, npos = r.ue(), r.ue()
        f["st_rps"].append((nneg, npos, [(r.ue(), r.u(1)) for _ in range(nneg + npos)]))
    f["long_term"] = r.u(1); f["temporal_mvp"] = r.u(1); f["strong_intra"] = r.u(1); f["vui"] = r.u(1); assert not f["vui"]
    f["ext"] = r.u(1); f["trailing"] = (r.u(r.left()), )
    return f


def parse_pps(n):
    r = Bits(unescape(n)[2:]); f = {}
    f["pps_id"] = r.ue(); f["sps_id"] = r.ue(); f["dep_slices"] = r.u(1); f["output_flag_present"] = r.u(1); f["extra_bits"] = r.u(3)
    f["sign_hiding"] = r.u(1); f["cabac_init_present"] = r.u(1); f["l0_m1"] = r.ue(); f["l1_m1"] = r.ue(); f["init_qp_m26"] = r.se()
    f["constrained_intra"] = r.u(1); f["tskip"] = r.u(1); f["cu_qp_delta"] = r.u(1)
    if f["cu_qp_delta"]:
        f["diff_cu_qp_delta_depth"] = r.ue()
    f["cb_off"] = r.se(); f["cr_off"] = r.se(); f["slice_chroma_off"] = r.u(1); f["wp"] = r.u(1); f["wbp"] = r.u(1)
    f["tq_bypass"] = r.u(1); f["tiles_enabled"] = r.u(1); f["wpp"] = r.u(1); assert not f["wpp"]
    if f["tiles_enabled"]:
        f["tile_columns"] = r.ue() + 1; f["tile_rows"] = r.ue() + 1; f["uniform_spacing"] = r.u(1)
        assert f["uniform_spacing"]
        f["lf_across_tiles"] = r.u(1)
    f["lf_across_slices"] = r.u(1); f["dbk_control"] = r.u(1)
    if f["dbk_control"]:
        f["dbk_override"] = r.u(1); f["dbk_disabled"] = r.u(1)
        if not f["dbk_disabled"]:
            f["beta_div2"] = r.se(); f["tc_div2"] = r.se()
    f["scaling_list"] = r.u(1); f["lists_mod"] = r.u(1); f["log2_par_merge_m2"] = r.ue(); f["sh_ext"] = r.u(1); f["ext"] = r.u(1)
    f["trailing"] = (r.u(r.left()), )
    return f


def parse_slice_header(n, sps, pps):
    """I slices of the reference's all-intra configuration only."""
    nt = (n[0] >> 1) & 63
    r = Bits(unescape(n)[2:]); f = {"nal_type": nt}
    f["first_slice"] = r.u(1)
    if 16 <= nt <= 23:
        f["no_output_prior"] = r.u(1)
    f["pps_id"] = r.ue()
    assert f["first_slice"] == 1
    f["slice_type"] = r.ue()
    if nt not in (19, 20):
        f["poc_lsb"] = r.u(sps["log2_poc_m4"] + 4)
        f["st_rps_sps_flag"] = r.u(1)
        if not f["st_rps_sps_flag"]:
            f["rps"] = (r.ue(), r.ue())      # num_negative, num_positive (inter_ref_pic_set_prediction_flag absent for idx 0)
    if sps["sao"]:
        f["sao_luma"] = r.u(1); f["sao_chroma"] = r.u(1)
    f["qp_delta"] = r.se()
    if pps.get("dbk_override"):
        f["dbk_override_flag"] = r.u(1)
    f["header_bits"] = r.p
    if pps["lf_across_slices"]:             # deblocking on (and / or SAO): slice_loop_filter_across_slices_enabled_flag
        f["slice_lf_across_slices"] = r.u(1)
    f["entry_points"] = []
    if pps.get("tiles_enabled"):
        n_entry = r.ue()
        if n_entry:
            bits = r.ue() + 1
            f["entry_points"] = [r.u(bits) + 1 for _ in range(n_entry)]
    assert r.u(1) == 1                      # byte_alignment()
    while r.p % 8:
        assert r.u(1) == 0
    f["data_byte_pos"] = 2 + r.p // 8       # in the unescaped NAL (2-byte NAL header)
    return f, r
