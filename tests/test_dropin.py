"""The reference's entry scripts import the hot path by these names (eval_interactive_davis.py:11-15,
interactive_gui.py:29-35, davis_processor.py:7-9); after dropin.install() they resolve to the engine."""
import subprocess
import sys
import textwrap
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_lines_resolve_to_engine(tmp_path):
    fake_ref = tmp_path / "MiVOS"
    (fake_ref / "model" / "s2m").mkdir(parents=True)
    (fake_ref / "model" / "s2m" / "s2m_network.py").write_text("deeplabv3plus_resnet50 = 'reference S2M'\n")
    (fake_ref / "model" / "s2s").mkdir(parents=True)
    (fake_ref / "model" / "s2s" / "other.py").write_text("thing = 'reference-only module'\n")
    (fake_ref / "util").mkdir()
    (fake_ref / "util" / "palette.py").write_text("pal_color_map = 'reference palette'\n")
    script = fake_ref / "entry.py"
    script.write_text(textwrap.dedent("""
        # the import block of eval_interactive_davis.py / interactive_gui.py / davis_processor.py
        from model.propagation.prop_net import PropagationNetwork
        from model.fusion_net import FusionNet
        from model.s2m.s2m_network import deeplabv3plus_resnet50 as S2M
        from model.s2s.other import thing
        from davis_processor import DAVISProcessor
        from inference_core import InferenceCore
        from model.aggregate import aggregate_wbg, aggregate_sbg
        from util.tensor_util import pad_divide_by, unpad, unpad_3dim, compute_multi_class_iou
        from util.palette import pal_color_map
        from generation.fusion_generator import FusionGenerator            # generate_fusion.py:16
        from model.fusion_model import FusionModel                         # train.py:14
        from dataset.davis_test_dataset import DAVISTestDataset            # eval_interactive_davis.py:14
        from dataset.yv_test_dataset import YouTubeVOSTestDataset
        from dataset.range_transform import im_normalization, inv_im_trans  # interact/interactive_utils.py:15
        import inspect
        assert DAVISTestDataset.__module__ == 'mivos_amd.dataset.davis_test_dataset' and YouTubeVOSTestDataset.__module__ == 'mivos_amd.dataset.yv_test_dataset'
        assert list(inspect.signature(DAVISTestDataset.__init__).parameters)[1:6] == ['root', 'imset', 'resolution', 'single_object', 'target_name']
        assert list(inspect.signature(YouTubeVOSTestDataset.__init__).parameters)[1:3] == ['data_root', 'split']
        assert FusionGenerator.__module__ == 'mivos_amd.generation.fusion_generator' and FusionModel.__module__ == 'mivos_amd.model.fusion_model'
        assert list(inspect.signature(FusionGenerator.__init__).parameters)[1:] == ['prop_net', 'images', 'mem_freq']
        assert list(inspect.signature(FusionGenerator.interact_mask).parameters)[1:] == ['mask', 'idx', 'left_limit', 'right_limit']
        assert list(inspect.signature(FusionModel.__init__).parameters)[1:] == ['para', 'logger', 'save_path', 'local_rank', 'world_size', 'distributed']
        assert list(inspect.signature(FusionModel.do_pass).parameters)[1:] == ['data', 'it']
        assert PropagationNetwork.__module__ == 'mivos_amd.model.propagation.prop_net'
        assert InferenceCore.__module__ == 'mivos_amd.inference_core' and thing == 'reference-only module'
        assert S2M.__module__ == 'mivos_amd.model.s2m.s2m_network' and DAVISProcessor.__module__ == 'mivos_amd.davis_processor'
        assert list(inspect.signature(S2M).parameters) == ['num_classes', 'output_stride', 'pretrained_backbone']
        assert list(inspect.signature(DAVISProcessor.__init__).parameters)[1:] == ['prop_net', 'fuse_net', 's2m_net', 'images', 'num_objects', 'device']
        sig = inspect.signature(InferenceCore.__init__)
        assert list(sig.parameters)[1:] == ['prop_net', 'fuse_net', 'images', 'num_objects', 'mem_profile', 'mem_freq', 'device']
        assert sig.parameters['mem_freq'].default == 5 and sig.parameters['mem_profile'].default == 0
        assert list(inspect.signature(InferenceCore.interact).parameters)[1:] == ['mask', 'idx', 'total_cb', 'step_cb']
        assert inspect.signature(PropagationNetwork.__init__).parameters['top_k'].default == 50
        for m in ('memorize', 'get_query_values', 'segment_with_query', 'get_W', 'get_attention'):
            assert hasattr(PropagationNetwork, m), m
        assert list(inspect.signature(FusionNet.forward).parameters)[1:] == ['im', 'seg1', 'seg2', 'attn', 'time']
        print('DROPIN_OK', pal_color_map)
    """))
    out = subprocess.run([sys.executable, "-m", "mivos_amd.dropin", str(script)], capture_output=True, text=True,
                         cwd=ROOT, timeout=240)
    assert "DROPIN_OK reference palette" in out.stdout, out.stdout + out.stderr
