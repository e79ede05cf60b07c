! ref_shielding.f90 -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
!
! Calls the reference's own self-shielding functions (get_H2_self_shielding, src/disk.f90:1887-1897; get_12CO_shielding,
! src/load_Visser_CO_selfshielding.f90:271-309) on the points read from standard input and prints their values, so that the
! host-side helpers of rac-2d_amd/cells.py can be checked against them (tests/golden/make_golden.py shielding).
! Input : lines "H2 N_H2 dv_turb" or "CO N_H2 N_12CO".   Output: one value per line, ES25.17E3.
program ref_shielding
  use disk, only: get_H2_self_shielding
  use load_Visser_CO_selfshielding, only: get_12CO_shielding
  implicit none
  character(len=2) :: what
  double precision :: a, b
  integer :: ios
  do
    read(*, *, iostat=ios) what, a, b
    if (ios .ne. 0) exit
    if (what .eq. 'H2') then
      write(*, '(ES25.17E3)') get_H2_self_shielding(a, b)
    else
      write(*, '(ES25.17E3)') get_12CO_shielding(a, b)
    end if
  end do
end program ref_shielding
