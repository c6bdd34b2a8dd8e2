"""SURVEY 8(f)4, wire side: the token layouts between an EnCodec-style codec and the models (valle2_amd/codec_io.py).
The codec here is a FAKE object with `encodec`'s call shapes (valle/models/encodec_pip.py:24-85) — no SEANet / RVQ arithmetic
exists in this repo and none is claimed (parity unpinned for the codec itself)."""
import sys
import types

import pytest
import torch

from valle2_amd import codec_io as CIO
from valle2_amd.config import ConfigValle


class FakeEncodecModel:
    """`encodec.EncodecModel`'s surface as EncodecPip uses it: encode -> [(codes (B, Q, T), scale)], decode([(codes, None)])
    -> (B, 1, T * hop), encoder(x) -> (B, C, T).  Codes are a reversible function of the samples so round trips can be checked."""
    sample_rate = 24000
    Q, hop = 8, 320

    def encode(self, x):                                   # (B, 1, T)
        b, _, t = x.shape
        frames = -(-t // self.hop)
        base = (x[:, 0, ::self.hop][:, :frames] * 100).round().long() % 1024          # (B, frames)
        codes = torch.stack([(base + 7 * q) % 1024 for q in range(self.Q)], dim=1)    # (B, Q, frames)
        half = frames // 2
        return [(codes[..., :half], None), (codes[..., half:], None)]                 # two chunks, as the real model may emit

    def decode(self, frames):
        codes = frames[0][0]                               # (B, Q, T)
        return codes[:, :1].float().repeat_interleave(self.hop, dim=-1) / 100          # (B, 1, T * hop)

    def encoder(self, x):
        return x[:, :, ::self.hop].repeat(1, 128, 1)


CFG = dict(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2)


def test_frame_bookkeeping():
    assert CIO.ENCODEC_FRAME_RATE == 75
    assert CIO.seconds_to_frames(3.0) == 225 and CIO.seconds_to_frames(30.0) == 2250      # BASELINE configs[4]
    assert CIO.seconds_to_frames(1 / 24000) == 1 and CIO.frames_to_seconds(75) == 1.0


def test_layouts_and_validation():
    cfg = ConfigValle(**CFG)
    g = torch.Generator().manual_seed(0)
    codes = torch.randint(0, 1024, (8, 37), generator=g)
    assert CIO.validate_codes(codes, cfg) is codes
    pc = CIO.to_prompt_codes(codes, cfg)
    assert tuple(pc.shape) == (37, 8) and pc.is_contiguous() and torch.equal(pc.T, codes)
    assert torch.equal(CIO.from_generated(pc, cfg), codes)
    CIO.validate_codes(codes[None].repeat(3, 1, 1), cfg, batched=True)
    with pytest.raises(ValueError, match='codebooks'):
        CIO.validate_codes(codes.T.contiguous(), cfg)                  # (T, Q) where (Q, T) is expected
    with pytest.raises(ValueError, match='int64'):
        CIO.validate_codes(codes.int(), cfg)
    with pytest.raises(ValueError, match='2-D'):
        CIO.validate_codes(codes[0], cfg)
    bad = codes.clone()
    bad[3, 5] = cfg.eos_token
    with pytest.raises(IndexError, match='BOS / EOS'):
        CIO.validate_codes(bad, cfg)
    with pytest.raises(ValueError, match='generated'):
        CIO.from_generated(codes, cfg)                                 # (Q, T) where (T, Q) is expected
    tokens = torch.randint(0, 256, (11,), generator=g)
    item = CIO.to_collate_item(codes, tokens, cfg)
    with pytest.raises(IndexError):
        CIO.to_collate_item(codes, tokens + 300, cfg)
    # the item is what both collate functions take (valle/collate.py): AR keeps the first codebook, NAR transposes to (t, Q)
    from valle2_amd.collate import ValleARCollate, ValleNARCollate
    other = CIO.to_collate_item(torch.randint(0, 1024, (8, 20), generator=g), tokens[:5], cfg)
    ar = ValleARCollate(cfg)([item, other])
    assert tuple(ar['codes'].shape) == (2, 38) and int(ar['codes'][0, 0]) == cfg.bos_token
    assert torch.equal(ar['codes'][0, 1:], codes[0]) and int(ar['target'][1, 20]) == cfg.eos_token
    nar = ValleNARCollate(cfg)([item, other])
    assert tuple(nar['codes'].shape) == (2, 37, 8) and torch.equal(nar['codes'][0], codes.T)


def test_encodec_pip_adapter_over_a_fake_codec():
    pip = CIO.EncodecPip(FakeEncodecModel())
    assert pip.sampling_rate == 24000
    audio = torch.rand(24000)                                          # 1 s -> 75 frames
    codes = pip.encode(audio)
    assert tuple(codes.shape) == (8, 75) and codes.dtype == torch.int64
    CIO.validate_codes(codes, ConfigValle(**CFG))
    batch = pip.batch_encode(torch.stack([audio, audio.flip(0)]))
    assert tuple(batch.shape) == (2, 8, 75) and torch.equal(batch[0], codes)
    wav = pip.decode(codes)
    assert tuple(wav.shape) == (75 * 320,)
    assert tuple(pip.batch_decode(batch).shape) == (2, 75 * 320)
    assert torch.equal(pip.encode_decode(audio), wav)
    assert tuple(pip.get_embedding(audio).shape) == (128, 75) and tuple(pip.batch_get_embedding(audio[None]).shape) == (1, 128, 75)
    with pytest.raises(AssertionError, match='1D audio'):
        pip.encode(audio[None])
    with pytest.raises(AssertionError, match='2D codes'):
        pip.decode(codes[None])


def test_model_dict_resolves_encodec_pip_lazily(monkeypatch):
    import valle2_amd
    assert valle2_amd.MODEL_DICT.keys() == ['EncodecPip', 'ValleAR', 'ValleNAR']
    valle2_amd.MODEL_DICT.pop('EncodecPip', None)
    monkeypatch.setitem(sys.modules, 'encodec', None)                  # "not installed"
    with pytest.raises(ImportError, match='encodec'):
        valle2_amd.get_model_class('EncodecPip')
    fake = types.ModuleType('encodec')

    class EncodecModel:
        @staticmethod
        def encodec_model_24khz():
            m = FakeEncodecModel()
            m.bandwidth = None
            m.set_target_bandwidth = lambda bw: setattr(m, 'bandwidth', bw)
            return m
    fake.EncodecModel = EncodecModel
    monkeypatch.setitem(sys.modules, 'encodec', fake)                  # "installed"
    cls = valle2_amd.get_model_class('EncodecPip')
    assert cls is CIO.EncodecPip
    import valle.models
    assert valle.models.EncodecPip is cls                              # `from valle.models import EncodecPip`, as the reference exports it
    pip = cls()                                                        # the reference's constructor: 24 kHz model at 6 kbps
    assert pip.model.bandwidth == 6.0 and pip.sampling_rate == 24000
    valle2_amd.MODEL_DICT.pop('EncodecPip', None)


@pytest.mark.gpu
def test_round_trip_audio_to_ar_to_nar_to_codec_layout():
    """waveform -> fake codec (Q, T) -> ValleAR.generate -> ValleNAR.generate -> (Q, Ty) -> fake decode; the pieces equal what
    the models give when called directly with the transposed tensors."""
    from valle2_amd import get_model_class, synth
    ar_cfg = ConfigValle(**CFG, dropout=0.0, norm='LayerNorm', num_beams=2, top_k=1, max_audio_len=12)
    nar_cfg = ConfigValle(**CFG, dropout=0.0, norm='AdaptiveLayerNorm')
    ar = get_model_class('ValleAR')(ar_cfg)
    ar.load_state_dict(synth.silence_eos(synth.make_state_dict(ar_cfg, 'ValleAR', seed=1), ar_cfg))
    nar = get_model_class('ValleNAR')(nar_cfg)
    nar.load_state_dict(synth.make_state_dict(nar_cfg, 'ValleNAR', seed=2))
    ar, nar = ar.to('cuda').eval(), nar.to('cuda').eval()
    pip = CIO.EncodecPip(FakeEncodecModel())
    g = torch.Generator().manual_seed(5)
    wav = torch.rand(320 * 9, generator=g)
    pt, tt = torch.randint(0, 256, (5,), generator=g), torch.randint(0, 256, (7,), generator=g)
    codes, audio = CIO.synthesize(ar, nar, pt, wav, tt, codec=pip, greedy_nar=True)
    assert tuple(codes.shape) == (8, 12) and codes.dtype == torch.int64 and tuple(audio.shape) == (12 * 320,)
    prompt = CIO.to_prompt_codes(pip.encode(wav), ar_cfg)
    assert not codes.is_cuda                         # host inputs: host results (modules._on_device), computed on the device
    first = ar.generate(pt.cuda(), prompt.cuda(), tt.cuda())
    assert first.is_cuda and torch.equal(codes[0], first.cpu())
    direct = nar.generate(pt.cuda(), prompt.cuda(), tt.cuda(), first, greedy=True)
    assert torch.equal(codes, direct.T.cpu())
    assert torch.equal(CIO.synthesize(ar, nar, pt, pip.encode(wav), tt, greedy_nar=True), codes)      # codes in, codes out
