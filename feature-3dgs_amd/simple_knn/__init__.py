"""simple_knn - MI355X-native drop-in for the reference's second native module (submodules/simple-knn).

    from simple_knn._C import distCUDA2          # scene/gaussian_model.py:20,146

The compute is hand-written HIP behind the C ABI of libf3dgs_hip.so (feature-3dgs_amd/csrc/knn.hip); there is no CPU
fallback: importing `simple_knn._C` without the built extension raises.
"""
