"""The codec side of the path, WIRE FORMAT ONLY (reference: valle/models/encodec_pip.py:24-85,
valle/collate.py:19-60, valle/models/valle_ar.py:95-97, valle/models/valle_nar.py:107-116).

EnCodec's own arithmetic (SEANet encoder / decoder, residual vector quantiser) is NOT here: it is a third-party package
(`encodec==0.1.1`, absent from this image) whose weights are downloaded, so there is neither an oracle nor a checkpoint to
pin a kernel to — that part of the row stays "parity unpinned" (DESIGN.md section 7).  What IS here is everything between
the codec's tensors and the models, so that a user of the reference finds the same objects:

  layouts      codec  `(Q, T)` int64  (EncodecPip.encode, encodec_pip.py:24-41; `(B, Q, T)` batched, :43-57)
               collate item  {'codes': (Q, T), 'tokens': (n,)}  ->  ValleARCollate / ValleNARCollate (collate.py)
               generate()    `prompt_codes (T, Q)`  (valle_ar.py:95-97, valle_nar.py:107-116)
               AR output `(Ty,)` first codebook, NAR output `(Ty, Q)`  ->  codec `(Q, Ty)` for EncodecPip.decode (:59-72)
  validation   Q = config.num_quantizers codebooks (8 at 6 kbps, valle/config.py:15-17), ids in [0, num_audio_tokens),
               int64, no BOS / EOS inside codec tensors
  bookkeeping  75 frames per second at 24 kHz (hop 320): seconds <-> frames
  EncodecPip   the reference's class name and methods as a thin adapter over an `encodec` model object; resolved lazily by
               `MODEL_DICT['EncodecPip']` when the package imports, ImportError otherwise (as before)
  synthesize   prompt (text + audio or codes) + target text -> AR first codebook -> NAR codebooks 2..Q -> codec tensor
               (-> waveform when a codec object is given): the joint inference BASELINE configs[4] times
"""
from __future__ import annotations

import torch
from torch import Tensor

ENCODEC_SAMPLE_RATE = 24000
ENCODEC_HOP = 320
ENCODEC_FRAME_RATE = ENCODEC_SAMPLE_RATE // ENCODEC_HOP          # 75 codec frames per second


def seconds_to_frames(seconds: float) -> int:
    """Frames EnCodec 24 kHz emits for `seconds` of audio (ceil: its encoder pads the last hop)."""
    return -(-int(round(seconds * ENCODEC_SAMPLE_RATE)) // ENCODEC_HOP)


def frames_to_seconds(frames: int) -> float:
    return frames * ENCODEC_HOP / ENCODEC_SAMPLE_RATE


def validate_codes(codes: Tensor, config, name: str = 'codes', batched: bool = False) -> Tensor:
    """`codes` in the codec layout `(Q, T)` (or `(B, Q, T)`): int64, Q == config.num_quantizers, every id in
    [0, config.num_audio_tokens).  Raises ValueError / IndexError (what nn.Embedding would raise later, said earlier)."""
    want = 3 if batched else 2
    if not isinstance(codes, Tensor) or codes.dim() != want:
        raise ValueError(f'{name}: expected a {want}-D tensor in the codec layout {"(B, Q, T)" if batched else "(Q, T)"}, '
                         f'got {tuple(codes.shape) if isinstance(codes, Tensor) else type(codes)}')
    if codes.dtype != torch.int64:
        raise ValueError(f'{name}: codec ids are int64, got {codes.dtype}')
    q = codes.shape[-2]
    if q != config.num_quantizers:
        raise ValueError(f'{name}: {q} codebooks, config.num_quantizers = {config.num_quantizers} '
                         f'(a (T, Q) tensor handed over where (Q, T) is expected?)')
    if codes.numel():
        lo, hi = int(codes.min()), int(codes.max())
        if lo < 0 or hi >= config.num_audio_tokens:
            raise IndexError(f'{name}: ids span [{lo}, {hi}], the codebooks hold {config.num_audio_tokens} entries '
                             f'(BOS / EOS never appear inside codec tensors)')
    return codes


def to_collate_item(codes_qt: Tensor, tokens: Tensor, config) -> dict:
    """One dataset item as the collate functions take it (valle/collate.py:27-33,48-49): codes stay `(Q, T)`."""
    validate_codes(codes_qt, config)
    if tokens.dim() != 1 or tokens.dtype != torch.int64:
        raise ValueError(f'tokens: expected 1-D int64 text ids, got {tuple(tokens.shape)} {tokens.dtype}')
    if tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= config.vocab_size):
        raise IndexError(f'tokens: ids outside [0, {config.vocab_size})')
    return {'codes': codes_qt, 'tokens': tokens}


def to_prompt_codes(codes_qt: Tensor, config) -> Tensor:
    """Codec `(Q, T)` -> the `prompt_codes (T, Q)` argument of ValleAR.generate / ValleNAR.generate."""
    return validate_codes(codes_qt, config, 'prompt codes').transpose(0, 1).contiguous()


def from_generated(nar_codes_tq: Tensor, config) -> Tensor:
    """ValleNAR.generate's `(Ty, Q)` -> codec `(Q, Ty)` for EncodecPip.decode (encodec_pip.py:59-72)."""
    if nar_codes_tq.dim() != 2 or nar_codes_tq.shape[1] != config.num_quantizers:
        raise ValueError(f'generated codes: expected (Ty, {config.num_quantizers}), got {tuple(nar_codes_tq.shape)}')
    return validate_codes(nar_codes_tq.transpose(0, 1).contiguous(), config, 'generated codes')


class EncodecPip:
    """valle/models/encodec_pip.py:5-131 — same constructor, properties and methods, over an `encodec` model object.
    `model=None` builds `EncodecModel.encodec_model_24khz()` at 6 kbps as the reference does (needs the `encodec` package
    and its downloaded weights); tests inject a stand-in.  This class does layout work only; the codec computes."""

    def __init__(self, model=None):
        if model is None:
            from encodec import EncodecModel                      # ImportError here = the package is not installed
            model = EncodecModel.encodec_model_24khz()
            model.set_target_bandwidth(6.0)
        self.model = model

    @property
    def sampling_rate(self) -> int:
        return self.model.sample_rate

    @staticmethod
    def _cat_frames(encoded_frames) -> Tensor:
        return torch.cat([encoded[0] for encoded in encoded_frames], dim=-1)            # (B, Q, T)

    @torch.inference_mode()
    def encode(self, audio: Tensor) -> Tensor:
        assert audio.dim() == 1, f'Expected 1D audio tensor, got {audio.dim()}D'
        return self._cat_frames(self.model.encode(audio[None, None]))[0]                # (Q, T)

    @torch.inference_mode()
    def batch_encode(self, audios: Tensor) -> Tensor:
        assert audios.dim() == 2, f'Expected 2D audio tensor, got {audios.dim()}D'
        return self._cat_frames(self.model.encode(audios[:, None]))                      # (B, Q, T)

    @torch.inference_mode()
    def decode(self, codes: Tensor) -> Tensor:
        assert codes.dim() == 2, f'Expected 2D codes tensor, got {codes.dim()}D'
        return self.model.decode([(codes[None], None)])[0, 0]                            # (T,)

    @torch.inference_mode()
    def batch_decode(self, codes: Tensor) -> Tensor:
        assert codes.dim() == 3, f'Expected 3D codes tensor, got {codes.dim()}D'
        return self.model.decode([(codes, None)])[:, 0]                                  # (B, T)

    @torch.inference_mode()
    def encode_decode(self, audio: Tensor) -> Tensor:
        return self.decode(self.encode(audio))

    @torch.inference_mode()
    def get_embedding(self, audio: Tensor) -> Tensor:
        assert audio.dim() == 1, f'Expected 1D audio tensor, got {audio.dim()}D'
        return self.model.encoder(audio[None, None])[0]                                  # (C, T)

    @torch.inference_mode()
    def batch_get_embedding(self, audios: Tensor) -> Tensor:
        assert audios.dim() == 2, f'Expected 2D audio tensor, got {audios.dim()}D'
        return self.model.encoder(audios[:, None])


@torch.inference_mode()
def synthesize(ar, nar, prompt_tokens: Tensor, prompt, target_tokens: Tensor, codec=None, greedy_nar: bool = False):
    """Joint AR -> NAR inference of one utterance in the codec's layouts.

    prompt: codec codes `(Q, T)` int64 of the acoustic prompt, or — with a `codec` — a 1-D waveform to encode first.
    Returns codes `(Q, Ty)` of the synthesised speech, or `(codes, waveform)` when a codec is given.
    AR: first codebook by ValleAR.generate (valle_ar.py:92-180); NAR: codebooks 2..Q by ValleNAR.generate
    (valle_nar.py:107-165) conditioned on the same prompt and the AR output."""
    cfg = ar.config
    if prompt.dtype.is_floating_point:
        if codec is None:
            raise ValueError('synthesize: a waveform prompt needs a codec to encode it')
        prompt = codec.encode(prompt)
    prompt_codes = to_prompt_codes(prompt, cfg)                                         # (T, Q)
    first = ar.generate(prompt_tokens, prompt_codes, target_tokens)                    # (Ty,) first codebook
    if first.numel() == 0:
        raise RuntimeError('synthesize: the AR model emitted EOS at its first step (no frames to refine)')
    dev = first.device
    codes_tq = nar.generate(prompt_tokens.to(dev), prompt_codes.to(dev), target_tokens.to(dev), first, greedy=greedy_nar)
    out = from_generated(codes_tq, cfg)                                                 # (Q, Ty)
    if codec is None:
        return out
    return out, codec.decode(out.to(_codec_device(codec, out.device)))


def _codec_device(codec, default):
    """Where the codec's model lives (its first parameter), `default` for a parameter-free stand-in."""
    params = getattr(getattr(codec, 'model', None), 'parameters', None)
    first = next(iter(params()), None) if callable(params) else None
    return default if first is None else first.device
