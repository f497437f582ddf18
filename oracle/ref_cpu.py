"""CPU oracle for the video-dqn Q-learning hot path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU fp32 restatement of the reference's algorithm for the path
BASELINE.json's ``north_star`` names.  It is *not* part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The
product path (``video_dqn_amd``) never routes through it.

Pinning status (see DESIGN.md "Oracle"):
  * Model wiring (``HabitatDQNMultiAction``): pinned.  ``tests/golden/make_golden.py`` imports
    the reference class from ``/root/reference/archs/HabitatDQNMultiAction.py`` in the build
    container and records its state_dict key set and forward outputs; this restatement must
    reproduce them (tests/test_oracle_golden.py).
  * ``process_batch`` (``/root/reference/train_q_network.py:126-181``) is a closure inside
    ``run_train`` and cannot be imported; the golden generator extracts that function's AST from
    the reference file at generation time and executes it, so the loss goldens come from the
    reference's own statements.  This file restates it independently.
  * ``torchvision.models.resnet18`` (torchvision==0.4.2, reference requirements.txt:140) is a
    third-party dependency that is absent from /root/reference and from this image.  Its
    published topology is restated in ``ResNet18`` below; that part is pinned only by
    construction (module/key names + shapes), there is no reference golden for it.
  * The reference ships no tests / golden vectors for this path.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# torchvision 0.4.2 ResNet-18 topology (third-party; published definition restated).
#   conv1 7x7/2 p3 no-bias -> bn1 -> relu -> maxpool 3/2/1 -> layer1..4 (2 BasicBlocks each,
#   planes 64/128/256/512, stride 2 on the first conv of the first block of layer2..4, 1x1/2
#   conv+BN downsample there) -> avgpool -> fc(512, 1000).
#   Call sites: archs/HabitatDQNMultiAction.py:11,30,33.
# ----------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 2, 1)
        self.layer2 = self._make_layer(128, 2, 2)
        self.layer3 = self._make_layer(256, 2, 2)
        self.layer4 = self._make_layer(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        # torchvision's default init (used when no pretrained file is available)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(self.avgpool(x), 1)
        return self.fc(x)


def resnet18(pretrained=False, **kw):
    """Stand-in for ``torchvision.models.resnet18``; no pretrained file exists offline."""
    return ResNet18(**kw)


# ----------------------------------------------------------------------------------------------
# archs/HabitatDQNMultiAction.py:8-54
# ----------------------------------------------------------------------------------------------
class HabitatDQNMultiAction(nn.Module):
    def __init__(self, action_dim, num_classes=5, extra_capacity=False, panorama=True,
                 num_frames=None):
        # archs/HabitatDQNMultiAction.py:9-19.  ``num_frames`` generalises the hard-coded 4
        # (SURVEY.md D3); None reproduces the reference exactly.
        super().__init__()
        self.resnet = resnet18(pretrained=True)
        self.extra_capacity = extra_capacity
        self.num_classes = num_classes
        self.action_dim = action_dim
        self.panorama = panorama
        if num_frames is None:
            num_frames = 4 if panorama else 1
        self.num_frames = num_frames
        if extra_capacity:  # :27-31
            self.features = nn.Sequential(*list(self.resnet.children())[:-2],
                                          nn.Conv2d(512, 64, (3, 3)), nn.ReLU(), nn.Flatten())
            self.top = nn.Sequential(nn.Linear(1600 * self.num_frames, 512), nn.ReLU(),
                                     nn.Linear(512, 256), nn.ReLU(),
                                     nn.Linear(256, action_dim * self.num_classes))
        else:  # :32-34
            self.features = nn.Sequential(*list(self.resnet.children())[:-1])
            self.top = nn.Linear(512 * self.num_frames, action_dim * self.num_classes)

    def set_train(self):  # :37-40
        self.train()
        if self.extra_capacity:
            self.resnet.eval()

    def forward(self, inp):  # :44-54
        if self.num_frames == 1 and len(inp.shape) == 4:
            inp = inp.unsqueeze(1)
        if inp.shape[1] != self.num_frames:
            raise Exception("bad shape")
        feats = [self.features(inp[:, i, ...]) for i in range(self.num_frames)]
        combined = torch.cat(feats, 1).squeeze()
        out = self.top(combined)
        return out.view((-1, self.num_classes, self.action_dim))


def build_model(config, num_frames=None):
    """train_q_network.py:36-47."""
    actions = 1 if (config.VALUE_LEARNING or config.ONE_ACTION) else 3
    return HabitatDQNMultiAction(actions, 5,
                                 extra_capacity=(config.ARCHITECTURE == "extra_capacity"),
                                 panorama=(config.PANORAMA or config.PREVIOUS_IMAGES),
                                 num_frames=num_frames)


def default_config(**over):
    """defaults.py:7-37 merged with configs/experiments/real_data/config.yml:1-11."""
    cfg = dict(PANORAMA=False, SEED=4, TRAIN_ON_GROUND_TRUTH=False, DATASET="dataset/data.feather",
               CLASS_LABEL="all", LOSS_CLIP="rect", ARCHITECTURE="extra_capacity",
               ONE_ACTION=False, REMOVE_BEFORE_REWARD=False, USE_INVERSE_ACTIONS=True,
               VALUE_LEARNING=False, PREVIOUS_IMAGES=False, GAMMA=0.99, BOOTSTRAP=False,
               LINEAR=False, LEARNING_RATE=1e-4, NUM_STEPS=300000, TARGET_UPDATE_INTERVAL=8000,
               CHECKPOINT_INTERVAL=25000, CONFIDENCE_REWARD=False, VISUALIZATION_DATA_ROOT="")
    cfg.update(over)
    return SimpleNamespace(**cfg)


# ----------------------------------------------------------------------------------------------
# train_q_network.py:126-181  process_batch (restated; closure variables made arguments)
# ----------------------------------------------------------------------------------------------
def process_batch(model, target_net, config, batch, compare_ground_truth=False, detail=None):
    before, after, act, rew, term, ground_truth, valid_mask = batch  # :127-129
    before_values = model(before)  # :131
    action_indices = act.view(-1, 1).repeat(1, 5)  # :134
    Q_b = before_values.gather(2, action_indices.unsqueeze(2)).squeeze()  # :137
    if not compare_ground_truth:
        after_values = target_net(after)  # :140
        model_after_values = model(after)  # :142
        best_actions = model_after_values.argmax(-1)  # :148
        Q_a = after_values.gather(2, best_actions.unsqueeze(2)).detach().squeeze()  # :155-156
        Q_a = Q_a * (1 - term.float())  # :160
        if config.LINEAR:
            learn_targets = rew.float() + (Q_a - 0.1)  # :162
        else:
            learn_targets = rew.float() + config.GAMMA * Q_a  # :164
        if config.LOSS_CLIP == "rect":
            learn_targets = torch.clamp(learn_targets, max=1, min=0)  # :166
        losses = 0.5 * (Q_b - learn_targets) ** 2  # :167
        if config.REMOVE_BEFORE_REWARD:
            losses = losses * valid_mask  # :169
        if detail is not None:
            detail.update(after_values=after_values.detach(), best_actions=best_actions,
                          Q_a=Q_a, learn_targets=learn_targets,
                          model_after_values=model_after_values.detach())
    else:
        if config.VALUE_LEARNING:  # :172-176
            mask = 1 - torch.isnan(ground_truth).int()
            gt = ground_truth.clone()
            gt[torch.isnan(ground_truth)] = 0
            losses = 0.5 * (Q_b * mask - gt.float()) ** 2
        else:
            losses = 0.5 * (Q_b - ground_truth.float()) ** 2  # :178
    if detail is not None:
        detail.update(before_values=before_values, Q_b=Q_b, losses=losses)
    return losses.mean()  # :180


def td_loss_from_q(q_before, q_after_online, q_after_target, act, rew, term, valid_mask, config):
    """The elementwise tail of process_batch (:134-180) on already-computed Q tensors.

    Returns (loss, dL/dq_before).  Used as the oracle of the fused TD kernel.
    """
    qb = q_before.detach().clone().requires_grad_(True)
    Q_b = qb.gather(2, act.view(-1, 1).repeat(1, qb.shape[1]).unsqueeze(2)).squeeze(2)
    best = q_after_online.argmax(-1)
    Q_a = q_after_target.gather(2, best.unsqueeze(2)).detach().squeeze(2)
    Q_a = Q_a * (1 - term.float())
    y = rew.float() + (Q_a - 0.1) if config.LINEAR else rew.float() + config.GAMMA * Q_a
    if config.LOSS_CLIP == "rect":
        y = torch.clamp(y, max=1, min=0)
    losses = 0.5 * (Q_b - y) ** 2
    if config.REMOVE_BEFORE_REWARD:
        losses = losses * valid_mask
    loss = losses.mean()
    loss.backward()
    return loss.detach(), qb.grad, best, y


# ----------------------------------------------------------------------------------------------
# train_q_network.py:119-124, 208-231  one optimisation step in the reference's order
# ----------------------------------------------------------------------------------------------
class Trainer:
    def __init__(self, config, state_dict=None, num_frames=None):
        self.config = config
        self.model = build_model(config, num_frames)  # :119
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        self.target_net = build_model(config, num_frames)  # :120
        self.target_net.load_state_dict(self.model.state_dict())  # :121
        self.target_net.eval()  # :122
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=config.LEARNING_RATE)  # :124
        self.sample_number = 0

    def step(self, batch, detail=None):
        cfg = self.config
        self.sample_number += 1  # :213
        if self.sample_number % cfg.TARGET_UPDATE_INTERVAL == 0:  # :215-216
            self.target_net.load_state_dict(self.model.state_dict())
        self.model.set_train()  # :221
        self.optimizer.zero_grad()  # :222
        loss = process_batch(self.model, self.target_net, cfg, batch,
                             compare_ground_truth=cfg.TRAIN_ON_GROUND_TRUTH, detail=detail)  # :223
        loss.backward()  # :226
        self.optimizer.step()  # :227
        return loss.item()


# ----------------------------------------------------------------------------------------------
# util/torch.py:5-12,26-36  normalisation constants; dataloaders/q_learning_real.py:82-98
# ----------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
DETECTION_THRESHOLDS = (0.9700177907943726, 0.9738382697105408, 0.9512060284614563,
                        0.7334915995597839, 0.7058018445968628)  # q_learning_real.py:15-18


def to_imgnet(im_uint8_nhwc):
    """util/torch.py:26-36 — uint8 HWC/NHWC -> normalised float NCHW."""
    x = im_uint8_nhwc.float()
    squeeze = x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
    x = (x / 255).permute(0, 3, 1, 2)
    x = x - torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    x = x / torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return x.squeeze() if squeeze else x


def adam_reference(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (wd=0, no amsgrad); call site train_q_network.py:124,227."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ----------------------------------------------------------------------------------------------
# archs/inverse_action2.py:45-100  the inverse-action model used to label actions of new video data
# (dataset/process_episodes_real.py:84-95,164-179: model.eval(); acts = model(be, ae)[1].argmax(1))
# ----------------------------------------------------------------------------------------------
class InverseActionModel(nn.Module):
    def __init__(self):
        super().__init__()
        trunk = resnet18()
        self.resnet18 = nn.Sequential(*list(trunk.children())[:-2])  # :50-52
        self.resnet18.eval()  # :55-57 frozen
        for p in self.resnet18.parameters():
            p.requires_grad = False
        self.conv1 = nn.Conv2d(1024, 256, kernel_size=1)  # :60
        self.conv2 = nn.Conv2d(256, 256, kernel_size=3)  # :61
        self.conv3 = nn.Conv2d(256, 64, kernel_size=3)  # :62
        self.dropout1 = nn.Dropout2d(0.5)  # :63
        self.dropout2 = nn.Dropout2d(0.5)  # :64
        self.fc1 = nn.Linear(64 * 3 * 3, 128)  # :65
        self.fc2 = nn.Linear(128, 3)  # :66
        self.fc_accuracy = nn.Linear(3, 3)  # :69

    def forward(self, k, k_plus_one):  # :72-100
        self.resnet18.eval()
        a = self.resnet18(k)
        b = self.resnet18(k_plus_one)
        x = torch.cat([a, b], dim=1)
        x = torch.relu(self.conv1(x))
        x = torch.relu(self.conv2(x))
        x = torch.relu(self.conv3(x))
        x = x.view(x.size(0), -1)
        x = torch.relu(self.fc1(x))
        x = self.dropout1(x)
        x = self.fc2(x)
        encoding = torch.softmax(x, dim=1)
        y = self.fc_accuracy(x)
        return encoding, y


# ----------------------------------------------------------------------------------------------
# train_inverse_model.py:30-83  the TRAINING script's own copy of the model (an extra ReLU after fc2, one dropout)
# and one optimisation step of its loop (:86-112): CrossEntropyLoss, Adam(lr, weight_decay), StepLR per epoch (:190-199)
# ----------------------------------------------------------------------------------------------
class InverseTrainModel(InverseActionModel):
    def forward(self, k, k_plus_one, dropout_mask=None):  # :59-83
        self.resnet18.eval()
        x = torch.cat([self.resnet18(k), self.resnet18(k_plus_one)], dim=1)
        x = torch.relu(self.conv1(x))
        x = torch.relu(self.conv2(x))
        x = torch.relu(self.conv3(x))
        x = x.view(x.size(0), -1)
        x = torch.relu(self.fc1(x))
        if dropout_mask is not None:  # the mask torch's Dropout2d(0.5) drew in the golden run, replayed
            x = x * dropout_mask * 2.0
        elif self.training:
            x = self.dropout1(x)
        x = torch.relu(self.fc2(x))
        return self.fc_accuracy(x)


def inverse_train_step(model, optimizer, be, ae, act, dropout_mask=None):
    """train_inverse_model.py:93-112 for one minibatch; returns (loss, y)."""
    model.train()
    optimizer.zero_grad()
    y = model(be, ae, dropout_mask)
    loss = nn.CrossEntropyLoss()(y, act)
    loss.backward()
    optimizer.step()
    return loss.item(), y.detach()
