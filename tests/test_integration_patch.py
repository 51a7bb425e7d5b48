"""integration/vkrenderer_present.patch must apply to the reference as it is.  The patched renderer
cannot be compiled in either container (no Vulkan / GLFW / glm / ImGui), so the check is `patch --dry-run` on a scratch
copy of the three files it touches, plus a few structural assertions on the patch text.  Runs only where the reference
checkout is present (this container); skipped on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "vkrenderer_present.patch")
REF = "/root/reference"
FILES = ("CMakeLists.txt", "src/vulkan/VkRenderer.cpp", "src/vulkan/VkRenderer.h")


def test_patch_names_the_seams():
    text = open(PATCH).read()
    for needle in ("m_rt.upload(", "m_rt.setCamera(&ubo)", "m_rt.render(", "VK_IMAGE_USAGE_TRANSFER_DST_BIT", "vkCmdCopyBufferToImage",
                   "VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL", "renderFrameOnHip", "RT_AMD_ROOT", '#include "raytracer.hpp"'):
        assert needle in text, needle
    assert "-    vkCmdDrawIndexed(" in text                      # the full-screen draw through frag.glsl is what goes away
    assert all(("+++ b/" + f) in text for f in FILES)


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference checkout and patch(1)")
def test_patch_applies_to_the_reference(tmp_path):
    for f in FILES:
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copyfile(os.path.join(REF, f), dst)
    dry = subprocess.run(["patch", "--dry-run", "-p1", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert dry.returncode == 0 and "FAILED" not in dry.stdout and "fuzz" not in dry.stdout, dry.stdout + dry.stderr
    real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert real.returncode == 0, real.stdout + real.stderr
    cpp = (tmp_path / "src/vulkan/VkRenderer.cpp").read_text()
    assert "vkCmdDrawIndexed(commandBuffer" not in cpp and cpp.count("void VkRenderer::renderFrameOnHip()") == 1
    assert cpp.count("{") == cpp.count("}")                     # still balanced
