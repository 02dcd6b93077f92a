"""oracle/ref_args.py -- TEST INFRASTRUCTURE.  The reference encoder's configuration for this path as command-line switches.

The configuration of the hot path (CTU 64, 4 depths, TU 4..32, intra TU depth 3, all-intra GOP 1, RDOQ + RDOQTS, transform skip + fast,
sign hiding, deblocking + SAO, main profile) is what /root/reference/encoder_intra_main.cfg sets; the same settings given as `--Key=value`
switches let the reference build (oracle/_ref/TAppEncoder_ref) run where /root/reference does not exist (the GPU box: bench.py's CPU
baseline).  tests/test_ref_args.py proves, in the container that has the reference, that a run with these switches and a run with the
reference's own cfg file produce the same decisions, reconstruction and bitstream."""


def reference_args(width, height, n_frames, qp, bit_depth=8, level="6.2", frame_rate=30, wavefront=0):
    a = {
        "Profile": "main" if bit_depth == 8 else "main10", "MaxCUWidth": 64, "MaxCUHeight": 64, "MaxPartitionDepth": 4,
        "QuadtreeTULog2MaxSize": 5, "QuadtreeTULog2MinSize": 2, "QuadtreeTUMaxDepthInter": 3, "QuadtreeTUMaxDepthIntra": 3,
        "IntraPeriod": 1, "DecodingRefreshType": 1, "GOPSize": 1, "ReWriteParamSetsFlag": 1,
        "FastSearch": 1, "SearchRange": 64, "HadamardME": 1, "FEN": 1, "FDM": 1,
        "QP": qp, "MaxDeltaQP": 0, "MaxCuDQPDepth": 0, "DeltaQpRD": 0, "RDOQ": 1, "RDOQTS": 1,
        "LoopFilterOffsetInPPS": 1, "LoopFilterDisable": 0, "LoopFilterBetaOffset_div2": 0, "LoopFilterTcOffset_div2": 0, "DeblockingFilterMetric": 0,
        "InputBitDepth": bit_depth, "InternalBitDepth": bit_depth, "SAO": 1, "AMP": 1, "TransformSkip": 1, "TransformSkipFast": 1, "SAOLcuBoundary": 0,
        "SliceMode": 0, "LFCrossSliceBoundaryFlag": 1, "PCMEnabledFlag": 0, "LFCrossTileBoundaryFlag": 1, "WaveFrontSynchro": int(wavefront),
        "ScalingList": 0, "TransquantBypassEnable": 0, "CUTransquantBypassFlagForce": 0,
        "InputChromaFormat": 420, "FrameRate": frame_rate, "FrameSkip": 0, "SourceWidth": width, "SourceHeight": height,
        "FramesToBeEncoded": n_frames, "Level": level,
    }
    return ["--%s=%s" % kv for kv in a.items()]
