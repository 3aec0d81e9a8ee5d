"""ORACLE (test infrastructure, not product code): CPU restatement of the tensor part of the reference's training-time frame pipeline,
AVE/dataloader.py:346-394 behind the PIL RandAugment, with the random draws as arguments:
  ToTensor (torchvision: uint8 HWC -> float CHW / 255), tensor_normalize (:470-485), random_resized_crop (transforms/video_transforms.py:
  529-561: crop + F.interpolate(bilinear, align_corners=False)), horizontal_flip (:152-186), RandomErasing 'pixel' cube mode
  (transforms/random_erasing.py:118-152: the same box in every frame, fresh noise per frame).
Pinned by tests/golden/video_aug.npz, produced by the reference's own functions (tests/golden/make_golden.py::video_aug_case) with their
draws recorded.  Only tests/ may import this."""
import torch


def video_aug(frames_u8, params, noise_box, mean, std, size=224):
    """frames_u8 [T, H, W, 3] uint8; params (i, j, h, w, flip, top, left, eh, ew); noise_box [T, 3, eh, ew] -> fp32 [3, T, size, size]."""
    i, j, h, w, flip, top, left, eh, ew = [int(x) for x in params]
    x = frames_u8.permute(0, 3, 1, 2).float().div(255)                       # ToTensor per frame, stacked: T C H W
    x = x.permute(0, 2, 3, 1)                                                # T H W C (:360)
    x = (x - torch.as_tensor(mean)) / torch.as_tensor(std)                   # tensor_normalize
    x = x.permute(3, 0, 1, 2)                                                # C T H W
    x = torch.nn.functional.interpolate(x[:, :, i:i + h, j:j + w], size=(size, size), mode="bilinear", align_corners=False)
    if flip:
        x = x.flip((-1))
    x = x.permute(1, 0, 2, 3).clone()                                        # T C H W
    if eh > 0:
        x[:, :, top:top + eh, left:left + ew] = torch.as_tensor(noise_box)
    return x.permute(1, 0, 2, 3).contiguous()
