"""Static description of the lifting networks: body-part groups, layer shapes, weight-key grammar.

This is the single host-side source of truth for

* which input joints feed which body-part branch and in which output slot the decoded joints
  land (reference: lib/model/rie.py:306-357 for the grouping, :415-431 for the reassembly),
* the tensor names/shapes of the reference ``state_dict`` (the weight ABI, SURVEY.md A.4;
  reference constructors lib/model/rie.py:13-63, 110-120, 140-157, 178-253, 443-494 and
  lib/model/embedding.py:4-13).

Nothing here computes anything; `ray3d_amd.modules` turns the entry list into an ``nn.Module``
parameter tree, `ray3d_amd.synth` fills it deterministically, and the C-ABI library has its own
C++ copy of the same grammar (csrc/r3d_plan.cpp) which the tests cross-check against this one.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

# Branch order is the order of the five local features in the fused tensor
# (reference lib/model/rie.py:371: cat(xTorso, xLArm, xRArm, xLLeg, xRLeg)).
BRANCHES = ("Torso", "LArm", "RArm", "LLeg", "RLeg")

# joints per branch, keyed by the number of input joints (rie.py:308-331 / :334-357; F=2 and F=3
# use the same joints, only the per-joint feature count differs).
GROUPS: Dict[int, Dict[str, Tuple[int, ...]]] = {
    17: {"Torso": (0, 7, 8, 9, 10), "LArm": (14, 15, 16), "RArm": (11, 12, 13),
         "LLeg": (1, 2, 3), "RLeg": (4, 5, 6)},
    15: {"Torso": (0, 1, 14), "LArm": (2, 3, 4), "RArm": (5, 6, 7),
         "LLeg": (8, 9, 10), "RLeg": (11, 12, 13)},
    14: {"Torso": (0, 7), "LArm": (8, 9, 10), "RArm": (11, 12, 13),
         "LLeg": (4, 5, 6), "RLeg": (1, 2, 3)},
}

# Output reassembly (rie.py:426-431): output joint slot s <- (branch, index inside that
# branch's decoded joints).  For J=14/15 this is NOT the inverse of GROUPS (quirk Q2).
_T, _LA, _RA, _LL, _RL = BRANCHES
OUTPUT_ORDER: Dict[int, Tuple[Tuple[str, int], ...]] = {
    17: ((_T, 0),) + tuple((_LL, i) for i in range(3)) + tuple((_RL, i) for i in range(3))
        + tuple((_T, i) for i in range(1, 5)) + tuple((_RA, i) for i in range(3))
        + tuple((_LA, i) for i in range(3)),
    15: ((_T, 0), (_T, 1)) + tuple((_LL, i) for i in range(3)) + tuple((_RL, i) for i in range(3))
        + tuple((_RA, i) for i in range(3)) + tuple((_LA, i) for i in range(3)) + ((_T, 2),),
    14: ((_T, 0),) + tuple((_LL, i) for i in range(3)) + tuple((_RL, i) for i in range(3))
        + tuple((_RA, i) for i in range(3)) + tuple((_LA, i) for i in range(3)) + ((_T, 1),),
}

MLP_HIDDEN = 1024        # FCBlock linear_size at every call site (rie.py:226,232,244-253,494)
EMBED_MID = 32           # Embedding mid_channels default (embedding.py:5)
BN_EPS = 1e-5            # nn.BatchNorm1d default
SLOPE_MAIN = 0.2         # LeakyReLU in TemporalBlock/Linear/FCBlock (rie.py:27,113,155)
SLOPE_EMBED = 0.01       # nn.LeakyReLU() default in Embedding (embedding.py:7)


@dataclass(frozen=True)
class LiftConfig:
    """Hyper-parameters of one lifting network (pos or trj)."""
    kind: str                       # "pos" (RIEModel) | "trj" (RIETrajectoryModel)
    num_joints: int                 # NUM_KPTS
    in_features: int                # INPUT_DIM (3 for rays, 2 for plain 2D)
    filter_widths: Tuple[int, ...]  # ARCHITECTURE
    channels: int = 256             # CHANNELS
    latent: int = 256               # LATENT_FEATURES_DIM
    stage: int = 3                  # STAGE (pos only; 1 => no FuseBlocks)
    extrinsic_dim: int = 2          # EXTRINSIC_DIM (0 when CAMERA_EMBDDING is False)
    embed_dim: int = 64             # EMBEDD_DIM   (0 when CAMERA_EMBDDING is False)
    causal: bool = False
    dense: bool = False
    optimize1f: bool = True         # not DISABLE_OPTIMIZATIONS
    bf16x3: bool = False            # not a reference key: model_config.get('BF16X3') - the large GEMMs on the bf16 matrix
                                    # cores through exact three-term splits (fp32-equivalent)

    def __post_init__(self):
        if self.kind not in ("pos", "trj"):
            raise ValueError("kind must be 'pos' or 'trj'")
        if self.num_joints not in GROUPS:
            raise ValueError("NUM_KPTS must be one of %s" % sorted(GROUPS))
        if self.in_features not in (2, 3):
            raise ValueError("INPUT_DIM must be 2 or 3")
        if not self.filter_widths or any(w != 3 for w in self.filter_widths):
            # every shipped cfg uses width-3 filters; the ternary-tree kernels rely on it
            raise NotImplementedError("only ARCHITECTURE made of 3s is supported, got %r"
                                      % (self.filter_widths,))
        if self.causal and self.optimize1f:
            # the reference itself cannot run this pair: rie.py:94 slices the residual with the
            # dilated causal_shift of :47 and the add at :97 fails on mismatched lengths
            raise NotImplementedError("CAUSAL=True needs DISABLE_OPTIMIZATIONS=True: with the strided "
                                      "(Optimize1f) convolutions the reference forward raises at rie.py:97")

    @property
    def residual_tap(self) -> int:
        """Which of a level's three input frames is the residual (rie.py:88-94): the centre one, or -
        CAUSAL with the dilated convolutions, res = x[:, :, pad+shift : T-pad+shift], shift == pad -
        the last one.  DISABLE_OPTIMIZATIONS alone changes nothing for an RF-long window: the dilated
        stack evaluates the same ternary tree as the strided one."""
        return 2 if self.causal else 1

    @property
    def dense_convs(self) -> bool:
        """DENSE=True swaps every level's dilated 3-tap convolution for a dense one of 2*pad+1 taps (rie.py:49-53) -
        but only in the un-optimised constructor branch: with the strided (Optimize1f) convolutions the flag is read
        and ignored (:54-55), exactly as here."""
        return self.dense and not self.optimize1f

    def level_taps(self, level: int) -> int:
        """Kernel size of layers_conv[2*(level-1)] (level >= 1): 3, or 2 * 3**level + 1 for the dense ablation
        (pad of that level = dilation = 3**level, rie.py:44-53)."""
        return 2 * 3 ** level + 1 if self.dense_convs else self.filter_widths[level]

    @property
    def camera_embedding(self) -> bool:
        return self.extrinsic_dim > 0 and self.embed_dim > 0

    @property
    def receptive_field(self) -> int:
        rf = 1
        for w in self.filter_widths:
            rf *= w
        return rf

    @property
    def current_frame(self) -> int:
        """Index of the frame used as 'current' (quirk Q1, rie.py:290,304): RF // in_features."""
        return self.receptive_field // self.in_features

    def branch_names(self) -> Tuple[str, ...]:
        return BRANCHES if self.kind == "pos" else ("",)

    def branch_joints(self, branch: str) -> Tuple[int, ...]:
        if self.kind == "trj":
            return tuple(range(self.num_joints))
        return GROUPS[self.num_joints][branch]

    def branch_in_channels(self, branch: str) -> int:
        return 3 * len(self.branch_joints(branch)) * self.in_features

    def decoder_in_dim(self) -> int:
        if self.kind == "trj":
            return 2 * self.latent + (self.embed_dim if self.camera_embedding else 0)
        n = 2 if self.stage == 1 else 3
        return n * self.latent + (self.embed_dim if self.camera_embedding else 0)


@dataclass(frozen=True)
class Entry:
    key: str
    shape: Tuple[int, ...]
    role: str          # conv_w | lin_w | bias | bn_weight | bn_bias | bn_mean | bn_var | bn_count
    fan_in: int = 0
    activated: bool = True   # is the layer followed by a LeakyReLU (used only by synth scaling)
    is_buffer: bool = False


def _bn(prefix: str, c: int) -> List[Entry]:
    return [
        Entry(prefix + ".weight", (c,), "bn_weight"),
        Entry(prefix + ".bias", (c,), "bn_bias"),
        Entry(prefix + ".running_mean", (c,), "bn_mean", is_buffer=True),
        Entry(prefix + ".running_var", (c,), "bn_var", is_buffer=True),
        Entry(prefix + ".num_batches_tracked", (), "bn_count", is_buffer=True),
    ]


def _linear(prefix: str, cin: int, cout: int, activated: bool = True) -> List[Entry]:
    return [
        Entry(prefix + ".weight", (cout, cin), "lin_w", fan_in=cin, activated=activated),
        Entry(prefix + ".bias", (cout,), "bias", fan_in=cin),
    ]


def temporal_block_entries(prefix: str, cin: int, cfg: LiftConfig) -> List[Entry]:
    """lib/model/rie.py:13-63."""
    c, w = cfg.channels, cfg.filter_widths
    out = [Entry(prefix + ".expand_conv.weight", (c, cin, w[0]), "conv_w", fan_in=cin * w[0])]
    out += _bn(prefix + ".expand_bn", c)
    for i in range(1, len(w)):
        a, b = 2 * (i - 1), 2 * (i - 1) + 1
        out.append(Entry("%s.layers_conv.%d.weight" % (prefix, a), (c, c, cfg.level_taps(i)), "conv_w",
                         fan_in=c * cfg.level_taps(i)))
        out += _bn("%s.layers_bn.%d" % (prefix, a), c)
        out.append(Entry("%s.layers_conv.%d.weight" % (prefix, b), (c, c, 1), "conv_w", fan_in=c))
        out += _bn("%s.layers_bn.%d" % (prefix, b), c)
    out.append(Entry(prefix + ".shrink.weight", (cfg.latent, c, 1), "conv_w", fan_in=c,
                     activated=False))
    out.append(Entry(prefix + ".shrink.bias", (cfg.latent,), "bias", fan_in=c))
    return out


def fc_block_entries(prefix: str, cin: int, cout: int, nblocks: int) -> List[Entry]:
    """lib/model/rie.py:138-157 (FCBlock) with lib/model/rie.py:108-120 (Linear) residual units."""
    h = MLP_HIDDEN
    out = _linear(prefix + ".fc_1", cin, h) + _bn(prefix + ".bn_1", h)
    for n in range(nblocks):
        p = "%s.layers.%d" % (prefix, n)
        out += _linear(p + ".w1", h, h) + _bn(p + ".batch_norm1", h)
        out += _linear(p + ".w2", h, h) + _bn(p + ".batch_norm2", h)
    out += _linear(prefix + ".fc_2", h, cout, activated=False)
    return out


def embedding_entries(prefix: str, cin: int, cout: int) -> List[Entry]:
    """lib/model/embedding.py:4-13."""
    return (_linear(prefix + ".w1", cin, EMBED_MID) + _bn(prefix + ".b1", EMBED_MID)
            + _linear(prefix + ".w2", EMBED_MID, cout) + _bn(prefix + ".b2", cout))


def state_entries(cfg: LiftConfig) -> List[Entry]:
    """Every tensor of the reference module's ``state_dict`` for this configuration."""
    jf = cfg.num_joints * cfg.in_features
    out: List[Entry] = []
    if cfg.kind == "pos":
        for b in BRANCHES:
            out += temporal_block_entries("LocalLayer_" + b, cfg.branch_in_channels(b), cfg)
        out += fc_block_entries("GlobalInfo", jf, cfg.latent, 2)
        if cfg.stage != 1:
            for i in range(5):
                out += fc_block_entries("FuseBlocks.%d" % i, 4 * cfg.latent, cfg.latent, 1)
        if cfg.camera_embedding:
            out += embedding_entries("embedder", cfg.extrinsic_dim, cfg.embed_dim)
        for b in BRANCHES:
            out += fc_block_entries("Integration_" + b, cfg.decoder_in_dim(),
                                    3 * len(cfg.branch_joints(b)), 1)
    else:
        out += temporal_block_entries("LocalLayer", cfg.branch_in_channels(""), cfg)
        out += fc_block_entries("GlobalInfo", jf, cfg.latent, 2)
        if cfg.camera_embedding:
            out += embedding_entries("embedder", cfg.extrinsic_dim, cfg.embed_dim)
        out += fc_block_entries("Integration", cfg.decoder_in_dim(), 3, 1)
    return out


def config_from_dicts(model_config: dict, kind: str) -> LiftConfig:
    """Read the reference's ``model_config`` keys (lib/model/__init__.py:11-46)."""
    if model_config["MODEL"] != "RIE":
        raise ValueError("Unrecognized mdoel {}".format(model_config["MODEL"]))
    if model_config["CAMERA_EMBDDING"]:
        ed, dd = int(model_config["EXTRINSIC_DIM"]), int(model_config["EMBEDD_DIM"])
    else:
        ed, dd = 0, 0
    widths = tuple(int(x) for x in str(model_config["ARCHITECTURE"]).split(","))
    return LiftConfig(
        kind=kind,
        num_joints=int(model_config["NUM_KPTS"]),
        in_features=int(model_config["INPUT_DIM"]),
        filter_widths=widths,
        channels=int(model_config["CHANNELS"]),
        latent=int(model_config["LATENT_FEATURES_DIM"]),
        stage=int(model_config["STAGE"]),
        extrinsic_dim=ed, embed_dim=dd,
        causal=bool(model_config["CAUSAL"]), dense=bool(model_config["DENSE"]),
        optimize1f=not bool(model_config["DISABLE_OPTIMIZATIONS"]),
        bf16x3=bool(model_config.get("BF16X3", False)),
    )


def default_model_config(**over) -> dict:
    """The model_config keys of cfg/cfg_ray3d_h36m_stage3.py:25-75 that the factory reads."""
    d = {
        "MODEL": "RIE", "TRAJECTORY_MODEL": True, "ARCHITECTURE": "3,3", "DROPOUT": 0.2,
        "NUM_FRAMES": 9, "CAUSAL": False, "CHANNELS": 256, "DENSE": False, "NUM_KPTS": 17,
        "INPUT_DIM": 3, "CAMERA_EMBDDING": True, "EXTRINSIC_DIM": 2, "EMBEDD_DIM": 64,
        "DISABLE_OPTIMIZATIONS": False, "STAGE": 3, "LATENT_FEATURES_DIM": 256,
    }
    d.update(over)
    if "ARCHITECTURE" in over and "NUM_FRAMES" not in over:
        rf = 1
        for x in str(d["ARCHITECTURE"]).split(","):
            rf *= int(x)
        d["NUM_FRAMES"] = rf
    return d
