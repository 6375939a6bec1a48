"""Mirror of renderer/pbr_materials.py: `PBRMaterials` is a parameter holder here — compute_tangent / apply_normal_map
(pbr_materials.py:58-124) run inside the fused shader kernel (csrc/shade.hip)."""
import torch


class PBRMaterials:
    def __init__(self, ambient_color=((1, 1, 1),), diffuse_color=((1, 1, 1),), specular_color=((1, 1, 1),), shininess=64,
                 normal_maps=None, device="cpu"):
        for n, c in (("ambient_color", ambient_color), ("diffuse_color", diffuse_color), ("specular_color", specular_color)):
            t = torch.as_tensor(c, dtype=torch.float32)
            if t.shape[-1] != 3:
                raise ValueError("Expected %s to have shape (N, 3); got %r" % (n, t.shape))        # pbr_materials.py:50-53
            setattr(self, n, t)
        self.shininess = float(shininess)
        if self.shininess != 0.0:
            raise NotImplementedError("HARP always builds PBRMaterials(shininess=0.0) (utils/visualize.py:259-263); other exponents "
                                      "are not implemented by the fused shader")
        self.normal_maps = normal_maps
        self.use_normal_map = normal_maps is not None
        self.device = device

    def clone(self):
        return PBRMaterials(self.ambient_color, self.diffuse_color, self.specular_color, self.shininess, self.normal_maps, self.device)
