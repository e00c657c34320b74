"""B1: encode every retrieved image AND its caption with CLIP -> the `mined.pth` dict the samplers read
(mirror of the reference's retrieval/extract_mined_feature.py:103-213, 252-280).

ref: extract_mined_feats_batch :103-213 (pass 1 captions -> encode_text -> L2-norm; pass 2 images -> encode_image ->
L2-norm; both in loader order), pre_extract_mined_fea :252-280 (bsz 1024, saves {caption_features, labels, filepath,
image_features}).  Both towers run on the HIP kernels; features are kept on the GPU and moved to the host once.
"""
import torch

from ..features import l2_normalize


@torch.no_grad()
def extract_mined_feats_batch(model, dataloader, tokenizer, device="cuda"):
    """dataloader yields (imgs [B,3,H,W], labels [B], captions list[str], filepaths list[str])."""
    model.eval()
    cap_feats, labels_lst, paths = [], [], []
    for imgs, labels, captions, filepaths in dataloader:          # pass 1: captions
        tok = tokenizer(list(captions)).to(device)
        cap_feats.append(l2_normalize(model.encode_text(tok)))
        labels_lst.append(labels.long())
        paths.extend(list(filepaths))
    img_feats = []
    for imgs, labels, captions, filepaths in dataloader:          # pass 2: images
        img_feats.append(l2_normalize(model.visual.forward_raw(imgs.to(device))))
    return {'caption_features': torch.cat(cap_feats).cpu(), 'labels': torch.cat(labels_lst).cpu(), 'filepath': paths,
            'image_features': torch.cat(img_feats).cpu()}


def pre_extract_mined_fea(model, dataloader, tokenizer, out_path, device="cuda"):
    feats = extract_mined_feats_batch(model, dataloader, tokenizer, device)
    torch.save(feats, out_path)
    return feats
