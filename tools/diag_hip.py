import sys, os, subprocess
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch_only":
    import torch; print("torch only:", torch.cuda.is_available(), torch.cuda.device_count())
elif mode == "torch_first":
    import torch; print("torch:", torch.cuda.is_available())
    from rust_raytrace_amd import _ffi; print("rtmi count:", _ffi.lib().rtmi_device_count())
    x = torch.zeros(4, device="cuda"); print(x.sum().item())
elif mode == "rtmi_first":
    from rust_raytrace_amd import _ffi; print("rtmi count:", _ffi.lib().rtmi_device_count())
    import torch; print("torch:", torch.cuda.is_available())
    os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
