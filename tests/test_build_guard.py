"""The build-time guard of csrc/check_resources.py (CPU): kernels whose pipelines count vector-memory operations must
not spill -- a spill is a scratch access the counted s_waitcnt vmcnt(N) do not know about."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("check_resources", os.path.join(ROOT, "unet_amd", "csrc", "check_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _remarks(scratch_by_kernel):
    lines = []
    for name, (scratch, spills) in scratch_by_kernel.items():
        lines += [f"conv.hip:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]",
                  "conv.hip:1:1: remark:     VGPRs: 200 [-Rpass-analysis=kernel-resource-usage]",
                  f"conv.hip:1:1: remark:     ScratchSize [bytes/lane]: {scratch} [-Rpass-analysis=kernel-resource-usage]",
                  "conv.hip:1:1: remark:     Dynamic Stack: False [-Rpass-analysis=kernel-resource-usage]",
                  f"conv.hip:1:1: remark:     VGPRs Spill: {spills} [-Rpass-analysis=kernel-resource-usage]"]
    return "\n".join(lines) + "\n"


def test_guard_accepts_clean_kernels_and_rejects_a_spill(tmp_path, capsys):
    mod = _load()
    counted = {f"_ZN12_GLOBAL__N_120conv_stream_kernel_tILi{i}EEEvNS_5ConvKEj": (0, 0) for i in range(4)}
    counted.update({f"conv_fwd2_kernel_{w}": (0, 0) for w in (64, 128, 192, 256)})
    counted["conv_wgrad2_kernel"] = (0, 0)
    counted["some_row_kernel"] = (64, 16)           # not a counted-wait kernel: may spill
    ok = tmp_path / "ok.remarks"
    ok.write_text(_remarks(counted))
    assert mod.main(str(ok)) == 0
    counted["_ZN12_GLOBAL__N_120conv_stream_kernel_tILi1EEEvNS_5ConvKEj"] = (176, 44)
    bad = tmp_path / "bad.remarks"
    bad.write_text(_remarks(counted))
    assert mod.main(str(bad)) == 1
    assert "conv_stream_kernel_tILi1" in capsys.readouterr().err
    # a remarks file without the kernels (flag dropped from the compile line) is an error too, not a silent pass
    empty = tmp_path / "empty.remarks"
    empty.write_text(_remarks({"some_row_kernel": (0, 0)}))
    assert mod.main(str(empty)) == 1


def test_the_real_build_left_a_clean_report():
    """The remarks of the conv.hip compile that produced the library in this tree, when present."""
    path = os.path.join(ROOT, "unet_amd", "csrc", "conv.remarks")
    if os.path.exists(path):
        assert _load().main(path) == 0
