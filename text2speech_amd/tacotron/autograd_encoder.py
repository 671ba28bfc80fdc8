"""Encoder part of the Tacotron-2 backward (BiLSTM BPTT, conv+BatchNorm stack, embedding)."""


def encoder_backward(bw, d_memory):
    """d_memory: [B][T_in][enc_dim] gradient w.r.t. the encoder output.  (stage 3: filled in below)"""
    return
