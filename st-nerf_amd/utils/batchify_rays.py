"""``layered_batchify_ray`` with the reference's signature (utils/batchify_rays.py:51-140)."""


def layered_batchify_ray(model, rays, labels, bboxes, chuncks=512 * 7, near_far=None, near_far_points=[],
                         density_threshold=0, bkgd_density_threshold=0):
    """utils/batchify_rays.py:51-140.  Fewer rays than one chunk: the model is called WITHOUT the
    thresholds (its defaults 1e-4 / 0 apply, :52-54).  Otherwise the reference loops over
    ``chuncks``-ray pieces on the host; here the pieces only define which row supplies the per-chunk
    boxes, and the kernels run over up to ``model.max_rays_per_launch`` rays at a time.

    Under an initialised torch.distributed group of more than one rank (one process per GPU; ``torchrun ... -m
    stnerf_amd.dropin demo/...`` sets that up) the chunks are dealt out to the ranks in turn and the whole 5-tuple is
    all-gathered, so every rank returns what the single-GPU call returns, bit for bit (stnerf_amd.parallel)."""
    N = rays.size(0)
    if N < chuncks:
        return model(rays, labels, bboxes, near_far=near_far, near_far_points=near_far_points)
    from stnerf_amd import parallel
    act = parallel.active_group(model)
    if act is not None:
        return parallel.render_rays_sharded(model, rays, chuncks, density_threshold, bkgd_density_threshold, act=act)
    return model.render_rays(rays, False, density_threshold, bkgd_density_threshold, ref_chunk=chuncks)
