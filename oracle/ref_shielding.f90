! ref_shielding.f90 -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
!
! Calls the reference's own self-shielding functions (get_H2_self_shielding, src/disk.f90:1887-1897; get_12CO_shielding,
! src/load_Visser_CO_selfshielding.f90:271-309) on the points read from standard input and prints their values, so that the
! host-side helpers of rac-2d_amd/cells.py can be checked against them (tests/golden/make_golden.py shielding).
! Input : lines "H2 N_H2 dv_turb", "CO N_H2 N_12CO" or "XR E_keV dust_depletion ratioDust2HnucNum GrainRadius_CGS" (sigma_Xray_Bethell,
!         src/load_Bethell_Xray.f90:70-96, the cross section calc_Xray_ionization_rate sums over, src/disk.f90:1969-2010).
! Output: one value per line, ES25.17E3.
program ref_shielding
  use disk, only: get_H2_self_shielding
  use load_Visser_CO_selfshielding, only: get_12CO_shielding
  use load_Bethell_Xray_cross, only: sigma_Xray_Bethell
  implicit none
  character(len=2) :: what
  character(len=256) :: line
  double precision :: a, b, c, d
  integer :: ios
  do
    read(*, '(A)', iostat=ios) line
    if (ios .ne. 0) exit
    read(line, *, iostat=ios) what
    if (ios .ne. 0) exit
    if (what .eq. 'XR') then
      read(line, *) what, a, b, c, d
      write(*, '(ES25.17E3)') sigma_Xray_Bethell(a, b, c, d)
      cycle
    end if
    read(line, *, iostat=ios) what, a, b
    if (ios .ne. 0) exit
    if (what .eq. 'H2') then
      write(*, '(ES25.17E3)') get_H2_self_shielding(a, b)
    else
      write(*, '(ES25.17E3)') get_12CO_shielding(a, b)
    end if
  end do
end program ref_shielding
