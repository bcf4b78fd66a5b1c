"""The compat headers (hobot_stereonet_amd/csrc/compat/include + rclcpp_stub) must declare every external
identifier and provide every header the REFERENCE's sources use from the closed dnn_node / libdnn packages and from
ROS 2.  Runs in the build container only: /root/reference does not exist on the GPU box.
What it reads from the reference: identifier names and #include lines, nothing else."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference/stereonet_infer"
COMPAT = os.path.join(ROOT, "hobot_stereonet_amd", "csrc", "compat")
INC_DIRS = [os.path.join(COMPAT, "include"), os.path.join(COMPAT, "rclcpp_stub")]

needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")


def ref_sources():
    return sorted(glob.glob(os.path.join(REF, "include", "*.h")) + glob.glob(os.path.join(REF, "src", "*.cpp")))


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"//[^\n]*", "", txt)


def compat_text():
    out = []
    for d in INC_DIRS:
        for dirpath, _, files in os.walk(d):
            for f in files:
                out.append(open(os.path.join(dirpath, f), errors="ignore").read())
    return strip_comments("\n".join(out))


@needs_reference
def test_every_external_identifier_is_declared():
    used = set()
    for f in ref_sources():
        txt = strip_comments(open(f, errors="ignore").read())
        used |= set(re.findall(r"\bhobot::dnn_node::([A-Za-z_]\w*)", txt))
        used |= set(re.findall(r"\b(hbDNN[A-Za-z_]\w*|hbSys[A-Za-z_]\w*|HB_[A-Z0-9_]+)\b", txt))
    assert {"DNNInput", "NV12PyramidInput", "DnnNode", "DNNTensor", "hbSysFlushMem", "HB_DNN_LAYOUT_NCHW"} <= used
    have = compat_text()
    missing = sorted(n for n in used if not re.search(r"\b%s\b" % re.escape(n), have))
    assert not missing, f"used by the reference but not declared under compat/include: {missing}"


@needs_reference
def test_every_external_header_resolves():
    own = {os.path.basename(f) for f in glob.glob(os.path.join(REF, "include", "*.h"))}
    wanted = set()
    for f in ref_sources():
        for inc in re.findall(r'#include\s+"([^"]+)"', strip_comments(open(f, errors="ignore").read())):
            if inc not in own and not inc.startswith("opencv2/"):
                wanted.add(inc)
    assert "dnn_node/util/image_proc.h" in wanted and "dnn_node/dnn_node.h" in wanted
    missing = sorted(i for i in wanted if not any(os.path.exists(os.path.join(d, i)) for d in INC_DIRS))
    assert not missing, f"headers the reference includes but compat does not provide: {missing}"


def test_members_the_reference_touches_exist():
    """Field / method names the reference reaches through the external types (from its call sites)."""
    have = compat_text()
    for name in ("sysMem", "virAddr", "phyAddr", "memSize", "properties", "validShape", "alignedShape", "dimensionSize",
                 "tensorLayout", "tensorType", "scaleData", "scaleLen", "msg_header", "output_tensors", "rt_stat",
                 "input_fps", "output_fps", "infer_time_ms", "fps_updated", "model_file", "model_task_type", "task_num",
                 "GetInputCount", "GetOutputCount", "GetInputTensorProperties", "GetDNNHandle", "GetModelInputSize",
                 "GetModel", "SetNodePara", "PostProcess", "dnn_node_para_ptr_"):
        assert re.search(r"\b%s\b" % name, have), name


def test_compat_headers_compile_standalone(tmp_path):
    """A translation unit that includes the external headers exactly as the reference's headers do (minus OpenCV,
    which this image lacks: stereonet_infer/include/stereonet_node.h:22-32, preprocess.h:21-39) and names the same
    types must compile."""
    src = tmp_path / "surface.cpp"
    src.write_text('''
#include "ai_msgs/msg/perception_targets.hpp"
#include "dnn_node/dnn_node.h"
#include "dnn_node/util/image_proc.h"
#include "dnn_node/dnn_node_data.h"
#include "hbm_img_msgs/msg/hbm_msg1080_p.hpp"
#include "sensor_msgs/msg/image.hpp"
using hobot::dnn_node::DNNInput;
using hobot::dnn_node::DNNTensor;
using hobot::dnn_node::Model;
using hobot::dnn_node::NV12PyramidInput;
using hobot::dnn_node::DnnNodeOutput;
int probe(hbDNNTensorProperties& p, hbSysMem& m) {
  std::shared_ptr<DNNInput> in = hobot::dnn_node::ImageProc::GetNV12PyramidFromNV12Img(nullptr, 2, 2, 2, 2);
  hbSysFlushMem(&m, HB_SYS_MEM_CACHE_CLEAN);
  hbSysFlushMem(&m, HB_SYS_MEM_CACHE_INVALIDATE);
  return p.tensorLayout == HB_DNN_LAYOUT_NCHW ? p.validShape.dimensionSize[3] : (p.tensorType == HB_DNN_TENSOR_TYPE_S8);
}
''')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include")]
    for d in INC_DIRS:
        cmd += ["-I", d]
    r = subprocess.run(cmd + [str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
