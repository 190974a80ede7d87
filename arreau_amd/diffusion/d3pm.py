"""D3PM buffers (reference: diffusion/d3pm.py:9-65).  The reverse step runs in the HIP library."""
import torch
import torch.nn as nn


class D3PM(nn.Module):
    def __init__(self, x0_model, n_T: int, num_classes: int = 10, forward_type="mask", hybrid_loss_coeff=0.001):
        super().__init__()
        if forward_type != "mask":
            raise NotImplementedError("the diffusion path uses the absorbing ('mask') chain (diffusion_loss.py:77-82)")
        self.n_T, self.num_classses, self.eps = n_T, num_classes, 1e-6
        self.hybrid_loss_coeff = hybrid_loss_coeff
        p_mask = 0.02  # transition_to_mask_state_prob, d3pm.py:34
        one = torch.zeros(num_classes, num_classes)
        one[:, -1] = p_mask
        one.diagonal().fill_(1 - p_mask)
        one[-1, -1] = 1
        mats = [one]
        for _ in range(1, n_T):
            mats.append(mats[-1] @ one)  # running products, d3pm.py:49-54
        self.register_buffer("q_one_step_transposed", one.t().unsqueeze(0).repeat(n_T, 1, 1).contiguous())
        self.register_buffer("q_mats", torch.stack(mats, 0))
