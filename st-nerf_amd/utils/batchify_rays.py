"""``layered_batchify_ray`` with the reference's signature (utils/batchify_rays.py:51-140)."""


def layered_batchify_ray(model, rays, labels, bboxes, chuncks=512 * 7, near_far=None, near_far_points=[],
                         density_threshold=0, bkgd_density_threshold=0):
    """utils/batchify_rays.py:51-140.  Fewer rays than one chunk: the model is called WITHOUT the
    thresholds (its defaults 1e-4 / 0 apply, :52-54).  Otherwise the reference loops over
    ``chuncks``-ray pieces on the host; here the pieces only define which row supplies the per-chunk
    boxes, and the kernels run over up to ``model.max_rays_per_launch`` rays at a time."""
    N = rays.size(0)
    if N < chuncks:
        return model(rays, labels, bboxes, near_far=near_far, near_far_points=near_far_points)
    return model.render_rays(rays, False, density_threshold, bkgd_density_threshold, ref_chunk=chuncks)
